/*
 * flake_oracle.h -- CPU restatement of libflake's prediction/entropy path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - fp64 LPC analysis (fo_lpc_*)      pinned bit-for-bit against the real
 *     reference lpc.c compiled into oracle/_ref (tests/test_oracle_vs_ref.py)
 *   - Rice search (fo_rice_*)           pinned against the real rice.c
 *   - Rice codeword emit                pinned against the real bitio.h
 *   - CRC-8/16                          pinned against the real crc.c
 *   - residual FIR, order-selection tree, stereo/wasted feeders, VBS split,
 *     frame assembly: optimize.c / encode.c / vbs.c need the CMake-generated
 *     config.h and are therefore NOT buildable here; these restatements are
 *     "parity unpinned" against a reference binary and are checked by FLAC
 *     format invariants instead (own decoder round trip, CRCs).
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference).
 */
#ifndef FLAKE_ORACLE_H
#define FLAKE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_MAX_ORDER      32   /* lpc.h:25 */
#define FO_MAX_PARTS      256  /* rice.h:34-35 */
#define FO_MAX_CH         8    /* encode.h:33 */
#define FO_MAX_BLOCK      65535

/* subframe types / channel modes, encode.h:37-46 */
enum { FO_SUB_CONSTANT = 0, FO_SUB_VERBATIM = 1, FO_SUB_FIXED = 8, FO_SUB_LPC = 32 };
enum { FO_CH_NOT_STEREO = 0, FO_CH_LEFT_RIGHT = 1, FO_CH_LEFT_SIDE = 8,
       FO_CH_RIGHT_SIDE = 9, FO_CH_MID_SIDE = 10 };
/* flake.h:38-58 */
enum { FO_OM_MAX = 0, FO_OM_EST, FO_OM_2LEVEL, FO_OM_4LEVEL, FO_OM_8LEVEL,
       FO_OM_SEARCH, FO_OM_LOG };
enum { FO_PRED_NONE = 0, FO_PRED_FIXED, FO_PRED_LEVINSON };
enum { FO_STEREO_INDEPENDENT = 0, FO_STEREO_ESTIMATE };

/* The slice of FlakeContext/FlakeEncodeParams the path reads (flake.h:59-197). */
typedef struct fo_params {
    int channels;
    int sample_rate;
    int bits_per_sample;
    int block_size;              /* params.block_size (maximum block) */
    int order_method;
    int stereo_method;
    int prediction_type;
    int min_prediction_order;
    int max_prediction_order;
    int min_partition_order;
    int max_partition_order;
    int variable_block_size;
    int allow_vbs;
    int lpc_precision;           /* always 15, encode.c:443 */
} fo_params;

/* What encode_residual() leaves in FlacSubframe (encode.h:52-63), flattened. */
typedef struct fo_subframe {
    int32_t type;
    int32_t type_code;
    int32_t order;
    int32_t shift;
    int32_t obits;
    int32_t wasted;
    int32_t rice_method;         /* 0 = RICE, 1 = RICE2 */
    int32_t porder;
    uint32_t est_bits;           /* return value of encode_residual() */
    int32_t ch_mode;             /* frame-level, replicated per subframe */
    int32_t rice_nbits;          /* exact length of the residual section in bits
                                    (2+4 header, params, codewords); 0 if none */
    int32_t reserved;
    int32_t coefs[FO_MAX_ORDER];
    int32_t rparams[FO_MAX_PARTS];
    int32_t warmup[FO_MAX_ORDER];   /* residual[0..order); [0] = the CONSTANT value */
} fo_subframe;

void fo_set_defaults(fo_params *p, int level);                 /* encode.c:158-266 */

/* ---- lpc.c ---- */
void fo_window_autocorr(const int32_t *smp, int n, int lag, double *autoc);
void fo_levinson(const double *autoc, int max_order, const double *ref,
                 double *lpc /* [32][32] row-major */);
int  fo_schur_order_est(const double *autoc, int max_order, double *lpc);
void fo_quantize_coefs(double *lpc_row, int order, int precision,
                       int32_t *out, int *shift);
int  fo_lpc_calc_coefs(const int32_t *smp, int n, int max_order, int precision,
                       int omethod, int32_t *coefs /* [32][32] */, int *shift /* [32] */);

/* ---- optimize.c ---- */
void fo_residual_fixed(int32_t *res, const int32_t *smp, int n, int order);
void fo_residual_lpc(int32_t *res, const int32_t *smp, int n, int order,
                     const int32_t *coefs, int shift);
/* encode_residual(): fills sf (except obits/wasted/ch_mode which are inputs),
 * writes res[n]; returns the reference's return value. */
int  fo_encode_residual(const fo_params *p, fo_subframe *sf,
                        const int32_t *smp, int32_t *res, int n);

/* ---- rice.c ---- */
int      fo_rice_best_k(uint64_t sum, int n);
uint32_t fo_rice_search(fo_subframe *sf, int pmin, int pmax,
                        const int32_t *res, int n, int pred_order);
uint32_t fo_subframe_bits(fo_subframe *sf, int pmin, int pmax, const int32_t *res,
                          int n, int pred_order, int bps, int precision, int lpc);

/* ---- encode.c feeders ---- */
int  fo_stereo_mode(const int32_t *left, const int32_t *right, int n);
/* copy_samples + channel_decorrelation + remove_wasted_bits.
 * smp is [channels][n]; sf[ch].obits/wasted/ch_mode are set. Returns ch_mode. */
int  fo_prepare_frame(const fo_params *p, const int32_t *pcm, int n,
                      int32_t *smp, fo_subframe *sf);

/* ---- emit ---- */
/* Residual section exactly as output_residual() writes it, MSB-first, starting
 * at bit 0 of out (zero-padded to a byte). Returns the number of bits, or -1
 * if cap_bytes would be exceeded. */
int64_t fo_residual_section_bits(const fo_subframe *sf, const int32_t *res, int n);
int64_t fo_emit_residual(const fo_subframe *sf, const int32_t *res, int n,
                         uint8_t *out, int64_t cap_bytes);

uint8_t  fo_crc8(const uint8_t *d, uint32_t len);
uint16_t fo_crc16(const uint8_t *d, uint32_t len);

/* encode_frame(): one FLAC frame. frame_number is ctx->frame_count at entry.
 * buf_size plays the role of the reference's frame buffer length.
 * sf_out (optional, [channels]) and res_out (optional, [channels][n]) receive
 * the per-subframe decisions. Returns bytes written or -1. */
int fo_encode_frame(const fo_params *p, uint32_t frame_number, const int32_t *pcm,
                    int n, uint8_t *out, int buf_size,
                    fo_subframe *sf_out, int32_t *res_out, int *was_verbatim);

/* vbs.c */
void fo_vbs_split(const int32_t *pcm, int channels, int block_size,
                  int *frames, int sizes[8]);
/* flake_encode_frame() minus MD5: handles VBS and the frame counter.
 * *frame_count is updated like ctx->frame_count. */
int fo_encode_block(const fo_params *p, uint32_t *frame_count, const int32_t *pcm,
                    int block_size, uint8_t *out, int buf_size);

/* Whole batch through the hot path only (prepare + encode_residual + exact
 * rice bit count), the unit of work bench.py times as cpu_baseline. */
int fo_encode_subframes_batch(const fo_params *p, const int32_t *pcm, int nframes,
                              int n, fo_subframe *sf /* [nframes*ch] */,
                              int32_t *res /* [nframes][ch][n] or NULL */,
                              uint8_t *bits /* [nframes*ch][slot_bytes] or NULL */,
                              int64_t slot_bytes);

#ifdef __cplusplus
}
#endif
#endif
