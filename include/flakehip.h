/*
 * flakehip.h -- C ABI of the MI355X (gfx950) prediction/entropy layer.
 *
 * This is the drop-in boundary: the internal seam of libflake that the HIP
 * layer replaces is
 *
 *     int encode_residual(FlacEncodeContext *ctx, int ch)      optimize.h:27
 *                                                              optimize.c:124-276
 * together with its feeder stages in encode_frame() (encode.c:932-942:
 * copy_samples, channel_decorrelation, remove_wasted_bits) and the residual
 * section of output_subframes() (encode.c:766-798 + bitio.h:120-141) --
 * batched over many independent frames.  Everything here is plain C: no HIP,
 * no C++ and no torch types cross the boundary.  Pointers documented as
 * "device" are hipMalloc'd addresses (e.g. a torch tensor's data_ptr()).
 *
 * Error convention follows libflake (negative int on failure, encode.c:925-992)
 * with distinct codes; nothing aborts or throws across the ABI.
 *
 * File:line citations are relative to the reference tree (/root/reference).
 */
#ifndef FLAKEHIP_H
#define FLAKEHIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define FHIP_API __attribute__((visibility("default")))
#else
#define FHIP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define FHIP_MAX_ORDER   32     /* MAX_LPC_ORDER, lpc.h:25 */
#define FHIP_MAX_PARTS   256    /* MAX_PARTITIONS, rice.h:34-35 */
#define FHIP_MAX_CH      8      /* FLAC_MAX_CH, encode.h:33 */
#define FHIP_MAX_LAGS    (FHIP_MAX_ORDER + 1)

/* largest block (the reference's limit, encode.h:35); blocks above 16384 take
 * streaming variants of K0 and K3 (correct, not tuned) */
#define FHIP_MAX_BLOCK   65535    /* FLAC's and libflake's limit (encode.c:288) */

enum {
    FHIP_OK            =  0,
    FHIP_E_GENERIC     = -1,    /* libflake's only code */
    FHIP_E_HIP         = -2,    /* a HIP runtime call failed; see fhip_last_error() */
    FHIP_E_UNSUPPORTED = -3,    /* valid libflake parameters this layer does not cover */
    FHIP_E_INVALID     = -4,    /* parameters flake_validate_params() would reject */
    FHIP_E_NOMEM       = -5
};

/* subframe types (encode.h:37-40) and channel modes (encode.h:42-46) */
enum { FHIP_SUB_CONSTANT = 0, FHIP_SUB_VERBATIM = 1, FHIP_SUB_FIXED = 8, FHIP_SUB_LPC = 32 };
enum { FHIP_CH_NOT_STEREO = 0, FHIP_CH_LEFT_RIGHT = 1, FHIP_CH_LEFT_SIDE = 8,
       FHIP_CH_RIGHT_SIDE = 9, FHIP_CH_MID_SIDE = 10 };

/*
 * The fields of FlakeContext / FlakeEncodeParams (flake.h:59-197) that the
 * path reads, with the same names and value ranges, plus lpc_precision
 * (FlacEncodeContext.lpc_precision, always 15: encode.c:443).
 */
typedef struct fhip_params {
    int channels;                /* 1..8 */
    int sample_rate;
    int bits_per_sample;         /* 4..32 */
    int block_size;              /* params.block_size: the largest block */
    int order_method;            /* FLAKE_ORDER_METHOD_* 0..6, flake.h:38-46 */
    int stereo_method;           /* 0 independent, 1 estimate, flake.h:48-51 */
    int prediction_type;         /* 0 none, 1 fixed, 2 levinson, flake.h:53-57 */
    int min_prediction_order;
    int max_prediction_order;
    int min_partition_order;
    int max_partition_order;
    int variable_block_size;
    int allow_vbs;
    int lpc_precision;
} fhip_params;

/*
 * What encode_residual() leaves in FlacSubframe (encode.h:52-63) and its
 * RiceContext (rice.h:41-46), one record per (frame, channel).
 */
typedef struct fhip_subframe_info {
    int32_t  type;               /* FHIP_SUB_* */
    int32_t  type_code;
    int32_t  order;
    int32_t  shift;
    int32_t  obits;
    int32_t  wasted;             /* FlacSubframe.wasted_bits */
    int32_t  rice_method;        /* RiceContext.method: 0 RICE, 1 RICE2 */
    int32_t  porder;             /* RiceContext.porder */
    uint32_t est_bits;           /* return value of encode_residual() */
    int32_t  ch_mode;            /* FlacFrame.ch_mode, replicated per channel */
    int32_t  rice_nbits;         /* exact bit length of the residual section as
                                    output_residual() writes it; 0 when the
                                    subframe has none; -1 when it does not fit
                                    the caller's slot (nothing written) */
    int32_t  reserved;
    int32_t  coefs[FHIP_MAX_ORDER];
    int32_t  rparams[FHIP_MAX_PARTS];   /* RiceContext.params[0 .. 2^porder) */
    int32_t  warmup[FHIP_MAX_ORDER];    /* residual[0 .. order): the warm-up samples the
                                           subframe header carries (encode.c:834-837,
                                           :851-854); [0] is the value of a CONSTANT
                                           subframe (encode.c:800-807) */
} fhip_subframe_info;

typedef struct fhip_ctx fhip_ctx;

/* ---- lifetime ------------------------------------------------------- */

/* Number of HIP devices, or a negative code. */
FHIP_API int fhip_device_count(void);

/*
 * Create a handle bound to one device.  Plays the role of the buffers
 * flake_encode_init() allocates (encode.c:378-472): device workspaces for
 * max_frames frames of p->block_size samples per channel, so that no call
 * below allocates.  Returns FHIP_E_INVALID where flake_validate_params()
 * (encode.c:268-373) returns -1.
 */
FHIP_API int fhip_create(fhip_ctx **out, int device, const fhip_params *p, int max_frames);
FHIP_API void fhip_destroy(fhip_ctx *ctx);                       /* flake_encode_close, encode.c:1010 */

/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL
 * returns to the handle's own stream. */
FHIP_API int fhip_set_stream(fhip_ctx *ctx, void *hip_stream);
FHIP_API int fhip_sync(fhip_ctx *ctx);

FHIP_API const char *fhip_strerror(int code);
FHIP_API const char *fhip_last_error(const fhip_ctx *ctx);
FHIP_API const char *fhip_version(void);                          /* flake_get_version, encode.c:1028 */

/* ---- the hot path --------------------------------------------------- */

/*
 * One batch of nframes consecutive blocks, all of block_size samples per
 * channel.  All pointers are DEVICE pointers.  pcm follows
 * flake_encode_frame()'s input contract (flake.h:229, encode.c:541-553):
 * channel-interleaved int32, sign-extended to bits_per_sample.
 * Outputs other than info may be NULL.  Asynchronous on the handle's stream.
 */
typedef struct fhip_batch {
    const int32_t      *pcm;          /* [nframes][block_size][channels] */
    int                 nframes;
    int                 block_size;
    fhip_subframe_info *info;         /* [nframes*channels] */
    int32_t            *residual;     /* [nframes][channels][block_size] FlacSubframe.residual */
    uint8_t            *rice_bits;    /* [nframes*channels][rice_slot_bytes]: the residual
                                         section of each subframe, MSB-first from bit 0 of
                                         its slot; slot bytes past the section are untouched */
    int64_t             rice_slot_bytes;   /* multiple of 4 */
    /* stage outputs for parity checks; NULL in production */
    int32_t            *samples;      /* [nframes][channels][block_size] FlacSubframe.samples
                                         after decorrelation and wasted-bits removal */
    double             *autoc;        /* [nframes*channels][FHIP_MAX_LAGS] compute_autocorr */
    /* whole frames assembled on the device (optional; needs rice_bits):
     * frame header + CRC-8, subframes, CRC-16 and the verbatim fallback,
     * encode.c:718-764, :800-917, :949-964 */
    uint8_t            *frames;       /* [nframes][frame_stride] */
    int64_t             frame_stride; /* multiple of 4, >= fhip_frame_stride() */
    int32_t            *frame_bytes;  /* [nframes] bytes written per frame */
    uint32_t            first_frame_number;   /* FlacEncodeContext.frame_count of frame 0; frame f
                                                 carries first + f (or first + f*block_size when
                                                 allow_vbs: encode.c:969-975) */
    const uint32_t     *frame_numbers;        /* optional [nframes]: explicit number per frame
                                                 (ragged VBS batches); overrides the rule above */
} fhip_batch;

/* Bytes per frame slot that hold any frame of block_size samples: its verbatim
 * size (encode.c:521-527) plus alignment slack. */
FHIP_API int64_t fhip_frame_stride(const fhip_params *p, int block_size);

FHIP_API int fhip_encode_subframes_dev(fhip_ctx *ctx, const fhip_batch *b);

/* Host batch -> the batch's frames back to back in host memory: what the loop around
 * flake_encode_frame() writes to the file (flake.c:633-637), for nframes blocks at once.
 * b->pcm is host memory; b->frame_bytes (host, [nframes]) receives the size of every frame
 * (encode_frame's return value, encode.c:976); b->frame_numbers / first_frame_number as in
 * fhip_encode_subframes; b->info (host) is optional; the other outputs are ignored.  Frames
 * are assembled, packed (exclusive scan of their sizes + copy) and only then brought over:
 * the transfer is the stream's bytes, not the frames' verbatim-size slots.  *out_bytes = the
 * bytes written to out (<= out_cap, else FHIP_E_INVALID and nothing is written). */
FHIP_API int fhip_encode_frames_packed(fhip_ctx *ctx, const fhip_batch *b, uint8_t *out,
                                       int64_t out_cap, int64_t *out_bytes);
/* The same in two steps, for a caller that runs chunks of a stream through several handles
 * side by side (one host thread each) and only knows where a chunk's frames go once the
 * chunks before it have reported their sizes: _begin uploads, encodes, packs and returns the
 * chunk's byte count (b->frame_bytes filled); _fetch brings the packed frames to `out`. */
FHIP_API int fhip_frames_packed_begin(fhip_ctx *ctx, const fhip_batch *b, int64_t *total_bytes);
FHIP_API int fhip_frames_packed_fetch(fhip_ctx *ctx, uint8_t *out, int64_t out_cap);
/* Optional first step of _begin, for the same caller: only the upload of b->pcm (returns when the copy is
 * done).  A following _begin for the same pcm / nframes / block_size skips its own upload.  Several handles
 * that upload at the same time share the link and finish together; taking turns (each chunk's upload behind
 * the one before it) lets every other phase -- kernels, the download of the packed frames -- run beside the
 * next chunk's upload. */
FHIP_API int fhip_frames_packed_upload(fhip_ctx *ctx, const fhip_batch *b);
/* _fetch without waiting for the copy (it runs on a stream of the handle's own, beside the handle's next
 * upload); `out` is complete once fhip_frames_packed_fetch_wait() has returned.  The handle's next _begin
 * orders itself behind a download still in flight. */
FHIP_API int fhip_frames_packed_fetch_async(fhip_ctx *ctx, uint8_t *out, int64_t out_cap);
FHIP_API int fhip_frames_packed_fetch_wait(fhip_ctx *ctx);

/* Page-locked host memory for the host-pointer entries above (libflake mallocs its frame buffer,
 * encode.c:453-454; a caller's PCM is whatever it read the file into, flake.c:622-630): copies from
 * and to pageable memory go through the runtime's bounce buffers at ~2/3 of the link's rate.
 *   fhip_host_alloc / _free        a page-locked buffer (NULL when the runtime refuses)
 *   fhip_host_register / _unregister   page-lock a range the caller owns, in place; the range must
 *                                  stay mapped until it is unregistered.  FHIP_E_HIP when refused
 *                                  (the copies then run as before).
 * None of them needs a handle; all are optional. */
FHIP_API void *fhip_host_alloc(size_t bytes);
FHIP_API void fhip_host_free(void *p);
FHIP_API int fhip_host_register(void *p, size_t bytes);
FHIP_API int fhip_host_unregister(void *p);

/* The same for a variable-block-size stream (encode_frame_vbs, vbs.c:85-119, per block):
 * nblocks blocks of block_size samples in host memory -> every block split by split_frame_v1
 * (vbs.c:36-83) on the device, the pieces encoded (one pass of the path per piece length,
 * fhip_encode_blocks_vbs_dev below), the frames packed in stream order and fetched.
 * Frames are numbered by their first sample (encode.c:969-975: allow_vbs), starting at
 * first_frame_number for the first sample of pcm.  block_bytes[b] = bytes of block b's frames
 * (the return value of flake_encode_frame for that block); block_frames[b] (optional) = how many
 * frames it became; *max_frame_bytes = the largest frame (encode.c:967); *next_frame_number =
 * the number the next block's first frame takes.  The handle must have been created with
 * variable_block_size and allow_vbs set and max_frames >= 8 * nblocks. */
FHIP_API int fhip_encode_blocks_vbs_packed(fhip_ctx *ctx, const int32_t *pcm, int nblocks,
                                           int block_size, uint32_t first_frame_number,
                                           uint8_t *out, int64_t out_cap, int32_t *block_bytes,
                                           int32_t *block_frames, int64_t *out_bytes,
                                           int32_t *max_frame_bytes, uint32_t *next_frame_number);

/* The same with the blocks and the stream DEVICE-RESIDENT, and no host synchronisation inside:
 * pcm (device) -> split_frame_v1 per block, the piece tables (a piece is k eighths of its block:
 * eight bins of equal length, counted and scanned on the device), the path once per bin with the
 * pieces encoded where they lie, the frames packed in stream order into out->packed (device).
 * Asynchronous on the handle's stream; the host learns nothing about the split unless it reads
 * the outputs back.  All pointers in fhip_vbs_out are device pointers:
 *   packed        the frames back to back (vbs.c:104-116 for every block, concatenated)
 *   packed_cap    its size in bytes; a stream that does not fit sets totals[3] and frames past
 *                 the end are not written
 *   frame_bytes   optional [8 * nblocks]: size of the stream's i-th frame, i < totals[0]
 *   block_bytes   optional [nblocks]: bytes of block b's frames (flake_encode_frame's return value)
 *   block_frames  optional [nblocks]: frames block b became (1 = left whole, vbs.c:100)
 *   totals        [4] int64: frames, bytes, largest frame (encode.c:967), flags -- bit 0: the stream was cut
 *                 (did not fit packed_cap), bit 1: some frame of the stream was not encoded (frame_bytes <= 0;
 *                 its bytes are missing from packed)
 * pcm must be 16-BYTE ALIGNED (the splitter and the feeder stage read the blocks where they lie with
 * 16-byte loads); a misaligned pointer is refused with FHIP_E_INVALID.
 * Frame numbers as in fhip_encode_blocks_vbs_packed.  The handle: variable_block_size and
 * allow_vbs set, max_frames >= 8 * nblocks.  Such a handle sizes its subframe-indexed workspaces
 * (autoc, coefs 4 KB, shift, fin, K0 records) for 20 frame slots per block -- 2.5 x max_frames frames --
 * in fhip_create, whether or not a variable-block-size batch is ever run on it: create handles that
 * only see uniform batches without variable_block_size. */
typedef struct fhip_vbs_out {
    uint8_t  *packed;
    int64_t   packed_cap;
    int32_t  *frame_bytes;
    int32_t  *block_bytes;
    int32_t  *block_frames;
    int64_t  *totals;
} fhip_vbs_out;
FHIP_API int fhip_encode_blocks_vbs_dev(fhip_ctx *ctx, const int32_t *pcm, int nblocks, int block_size,
                                        uint32_t first_frame_number, const fhip_vbs_out *out);

/* Optional hint for a caller that streams batch after batch through one handle
 * (flake.c:622-663 calls flake_encode_frame block after block): start the feeder
 * stage of the NEXT batch -- copy_samples + channel_decorrelation +
 * remove_wasted_bits, encode.c:541-694 -- now, on a stream of the handle's own,
 * so that it runs beside the autocorrelation of the batch in flight (K0 is
 * HBM-bound, K1 is not).  Only next->pcm, nframes and block_size are read.
 * Contract: next->pcm (device memory) already holds the samples when this is
 * called -- it is NOT ordered after work queued on the handle's stream -- and
 * stays unchanged until the fhip_encode_subframes_dev() call for the same
 * pcm / nframes / block_size, which must be the next encode call on this handle,
 * has completed.  That call then skips its own feeder stage; its results are
 * identical with or without the hint.  A hint that the next call does not match
 * (or a batch that asks for `samples` / `autoc`) is dropped without effect. */
FHIP_API int fhip_prepare_ahead(fhip_ctx *ctx, const fhip_batch *next);

/* Same with HOST pointers: copies in, runs, copies out, synchronises.  When
 * `frames` is requested, `info` and `rice_bits` may be NULL (the frames are
 * complete; the sections then live only in a device workspace of
 * rice_slot_bytes per subframe, which must still be given). */
FHIP_API int fhip_encode_subframes(fhip_ctx *ctx, const fhip_batch *b_host);

/* ---- stage entry points (HOST pointers, synchronous) ---------------- */
/* Each mirrors one reference function over a batch, for stage-level parity. */

/* lpc_calc_coefs(), lpc.c:224-257, for nsub blocks of n samples.
 * coefs [nsub][32][32], shift [nsub][32] (rows the reference leaves
 * unwritten are zero), opt_order [nsub]; autoc [nsub][33] optional. */
FHIP_API int fhip_lpc_calc_coefs(fhip_ctx *ctx, const int32_t *samples, int nsub, int n,
                        int max_order, int precision, int omethod,
                        int32_t *coefs, int32_t *shift, int32_t *opt_order,
                        double *autoc);

/* encode_residual(), optimize.c:124-276, on prepared samples [nsub][n] with
 * info[s].obits set by the caller; uses the handle's params. */
FHIP_API int fhip_encode_residual(fhip_ctx *ctx, const int32_t *samples, int nsub, int n,
                         fhip_subframe_info *info, int32_t *residual,
                         uint8_t *rice_bits, int64_t rice_slot_bytes);

/* The bits[] table of encode_residual()'s LPC order searches (optimize.c:201-261: 2/4/8-LEVEL,
 * SEARCH, LOG) on prepared samples [nsub][n]: bits[s][order - 1] = the size estimate the
 * reference computes for that order (encode_residual_lpc + calc_rice_params_lpc, optimize.c:207-212,
 * :228-233, :250-255) for every order the handle's method visits, 0xFFFFFFFF for the others and
 * for a constant block.  info[s].obits as for fhip_encode_residual; bits 8..15 of
 * info[s].reserved may carry 1 + m with |x| < 2^m for every sample (what the feeder stage
 * records; 0 = unknown), which lets 16-bit blocks take the packed FIRs.  FHIP_E_UNSUPPORTED where
 * the search runs inside the encode kernel (other block sizes / methods).  bits: [nsub][32]. */
FHIP_API int fhip_order_search_bits(fhip_ctx *ctx, const int32_t *samples, int nsub, int n,
                           const fhip_subframe_info *info, uint32_t *bits);

/* calc_rice_params_lpc() / calc_rice_params_fixed(), rice.c:173-187, plus the
 * residual section of output_residual() (encode.c:766-798) on GIVEN residuals
 * [nsub][n]: fills info[].rice_method/porder/rparams/est_bits/rice_nbits. */
FHIP_API int fhip_calc_rice_params(fhip_ctx *ctx, const int32_t *residual, int nsub, int n,
                                   int pred_order, int lpc, int bps, int pmin, int pmax,
                                   fhip_subframe_info *info, uint8_t *rice_bits,
                                   int64_t rice_slot_bytes);

/* split_frame_v1(), vbs.c:36-83, for nblocks blocks of block_size samples per
 * channel (block_size a multiple of 8, >= 128): frames [nblocks] and
 * sizes [nblocks][8] exactly as the reference computes them. */
FHIP_API int fhip_vbs_split(fhip_ctx *ctx, const int32_t *pcm, int nblocks, int block_size,
                            int32_t *frames, int32_t *sizes);

/* copy_samples + channel_decorrelation + remove_wasted_bits,
 * encode.c:541-694, for nframes blocks; fills info[].obits/wasted/ch_mode. */
FHIP_API int fhip_prepare_frames(fhip_ctx *ctx, const int32_t *pcm, int nframes, int n,
                        int32_t *samples, fhip_subframe_info *info);

/* ---- measurement ---------------------------------------------------- */

/* With profiling on, every kernel launch of the hot path is bracketed by
 * hipEvents on the launch stream. */
FHIP_API int fhip_set_profiling(fhip_ctx *ctx, int on);
/* After fhip_sync(): accumulated milliseconds and launch counts per kernel
 * since the last reset; returns the number of kernels (<= cap). */
FHIP_API int fhip_get_kernel_times(fhip_ctx *ctx, const char **names, double *ms, int *launches,
                          int cap, int reset);

#ifdef __cplusplus
}
#endif
#endif /* FLAKEHIP_H */
