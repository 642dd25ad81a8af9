// kernels.h -- launch interface between the C ABI (api.hip) and the gfx950
// kernels (k0_prepare.hip ... k4_assemble.hip).  Internal to libflakehip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flakehip.h"

namespace fhip {

// Blocks up to this size keep a subframe's samples in LDS / registers (K3's k_encode and
// k_encode_pow2); longer ones, up to FHIP_MAX_BLOCK, stream them (k_encode_big).
constexpr int FHIP_MAX_RESIDENT_BLOCK = 16384;

// Everything a launch needs that does not change within a batch.
struct EncodeArgs {
    fhip_params p;
    int n;            // block size of this batch
    int nsub;         // subframes in this batch
};

// K0: copy_samples + channel_decorrelation + remove_wasted_bits
// (encode.c:541-694).  pcm [nframes][n][ch] -> smp [nframes][ch][n].
// decide_only: write obits / wasted / ch_mode to info[] and leave smp to the fused K1.
// allow_narrow: a channel whose samples all fit 16 bits is stored as int16[n] at the
// start of its row and flagged in info.reserved (see narrow_rows_ok).
// frame_src (optional): frame f's PCM starts at pcm + frame_src[f] (int32 units) instead of
// pcm + f*n*ch -- the pieces of a variable-block-size batch are encoded where they lie.
// dev_frames / dev_sub (optional, every launch_* below): the launch's real frame / subframe count
// lives on the device; nframes / nsub then size the grid (the bin's capacity) and workgroups past
// the count leave at once (device_util.h: dev_count).
hipError_t launch_prepare(hipStream_t st, const fhip_params &p, const int32_t *pcm,
                          int nframes, int n, int32_t *smp, fhip_subframe_info *info,
                          bool decide_only = false, bool allow_narrow = false,
                          const long long *frame_src = nullptr, const int32_t *dev_frames = nullptr);

// True when K0, K1 and K3 all handle 16-bit sample rows for such a batch; the
// caller then passes allow_narrow / narrow_ok to the three launches of the batch.
// (wave_typed_k1: the caller runs the wave-typed K1 whatever the count -- launch_autocorr_bins)
bool narrow_rows_ok(const fhip_params &p, int nsub, int n, bool lpc_path, bool wave_typed_k1 = false);

// True when K1 will also do K0's apply stage for such a batch (stereo, whole
// tiles, the wave-typed kernel): launch_prepare(decide_only) + launch_autocorr(pcm).
bool autocorr_fuses_prepare(const fhip_params &p, int nsub, int n);

// K1: apply_welch_window + compute_autocorr (lpc.c:28-71).
// smp [nsub][n] -> autoc [nsub][FHIP_MAX_LAGS].
// With pcm_fused (stereo PCM) the producers read it instead of smp, apply info[]'s
// channel mode / wasted bits and write smp_out.
// With lpc_out (see autocorr_does_lpc) the kernel also does K2's work for its
// subframes and launch_lpc is not needed.
// tile_ctr (optional): one arrival counter per tile of 32 subframes, zeroed once (fhip_create) -- with it a
// lag-split launch (small batches: two workgroups per tile) still runs K2 as its tail, in whichever of a tile's
// two workgroups arrives second (the counters run on, their parity tells the arrivals apart).
struct autocorr_lpc_out { int precision, omethod; int32_t *coefs, *shift, *opt_order, *fin; int32_t *tile_ctr = nullptr; };
bool autocorr_does_lpc(int nsub, int n, int max_order, bool have_tile_counters = false);
// True when the wave-typed K1 (k_autocorr_wt) serves such a batch.
bool autocorr_is_wave_typed(int nsub, int n, int max_order);
hipError_t launch_autocorr(hipStream_t st, const int32_t *smp, int nsub, int n,
                           int max_order, double *autoc, const int32_t *pcm_fused = nullptr,
                           int32_t *smp_out = nullptr, const fhip_subframe_info *info = nullptr,
                           const autocorr_lpc_out *lpc_out = nullptr, bool narrow_ok = false,
                           const int32_t *dev_sub = nullptr, int nsub_hint = 0);

// K2: compute_lpc_coefs / _est + quantize_lpc_coefs (lpc.c:77-257).
// coefs [nsub][32][32], shift [nsub][32], opt_order [nsub].
// fin [nsub][FIN_STRIDE]: for the MAX/EST order methods the one row the
// reference quantises, compact: coefs[0..32), shift, order (prefetched by K3).
constexpr int FIN_STRIDE = 72;      // int32 per row: 32 coefs, shift, order, sum|coef|, 1 spare,
constexpr int FIN_DBL = 36;         // ... the first 16 coefficients once more as doubles (zero past the order:
                                    //     K3 reads them by scalar loads, they never touch a vector register),
constexpr int FIN_PAIRS = 68;       // ... and the first 8 as four int16 pairs (lo: tap 2j+2, hi: tap 2j+1)
hipError_t launch_lpc(hipStream_t st, const double *autoc, int nsub, int max_order,
                      int precision, int omethod, int32_t *coefs, int32_t *shift,
                      int32_t *opt_order, int32_t *fin, const int32_t *dev_sub = nullptr);

// prep: the records K0 filled (obits, wasted, ch_mode, the 16-bit-row flag) -- the handle's
// own buffer, never info[] itself (the kernels' pointers to the two are __restrict__); K3
// copies K0's fields from there into info[].
// K3: encode_residual (optimize.c:124-276) incl. the Rice search (rice.c) and,
// when bits != NULL, the residual section of output_residual (encode.c:766-798).
hipError_t launch_encode(hipStream_t st, const fhip_params &p, const int32_t *smp,
                         int nsub, int n, const int32_t *coefs, const int32_t *shift,
                         const int32_t *opt_order, const int32_t *fin,
                         fhip_subframe_info *info,
                         int32_t *residual, uint8_t *bits, int64_t slot_bytes,
                         int raw_order = -1, int raw_lpc = 0, bool narrow_ok = false,
                         const fhip_subframe_info *prep = nullptr, bool order_known = false,
                         const int32_t *dev_sub = nullptr);

// K3-S: the LPC order searches (order methods 2..6, optimize.c:201-261) for block sizes it
// supports: bits[order] for every order the method can visit, the method's walk over that
// table, and the winner left as K2 leaves the single row of MAX / EST (opt_order[s], fin[s]);
// launch_encode(..., order_known = true) then encodes it with the lean instance.
bool order_search_supported(const fhip_params &p, int n);
hipError_t launch_order_search(hipStream_t st, const fhip_params &p, const int32_t *smp, int nsub,
                               int n, const int32_t *coefs, const int32_t *shift,
                               int32_t *opt_order, int32_t *fin, const fhip_subframe_info *prep,
                               bool narrow_ok, const int32_t *dev_sub = nullptr,
                               uint32_t *table_out = nullptr);

struct MultiBin;
// The thinly filled bins of a small ragged batch: several bins in one search launch and one K3 launch.
// order_search_group: >= 0, equal for bins that can share a launch (-1: launch_order_search for that bin alone);
// encode_group: 256 or -1 likewise.  mb: units = subframes, wg0 from the capacities, cnt = live subframes per bin, unit0 =
// the bin's first subframe, smp_off, narrow (+ slot, bits_off for K3); the workspaces whole.
int order_search_group(const fhip_params &p, int n);
hipError_t launch_order_search_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *smp,
                                    const int32_t *coefs, const int32_t *shift, int32_t *opt_order, int32_t *fin,
                                    const fhip_subframe_info *prep);
int encode_group(const fhip_params &p, int n, bool order_known);
hipError_t launch_encode_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *smp,
                              const int32_t *coefs, const int32_t *shift, const int32_t *opt_order, const int32_t *fin,
                              fhip_subframe_info *info, const fhip_subframe_info *prep, uint8_t *bits);

// K4: whole frames on the device (encode.c:718-764, :800-917, :949-964):
// frames [nframes][frame_stride] bytes, frame_bytes [nframes].
hipError_t launch_assemble(hipStream_t st, const fhip_params &p, const int32_t *pcm, int nframes,
                           int n, const fhip_subframe_info *info, const uint8_t *rice,
                           int64_t slot_bytes, uint8_t *frames, int64_t frame_stride,
                           int32_t *frame_bytes, uint32_t number_base, uint32_t number_step,
                           const uint32_t *numbers = nullptr, const long long *frame_src = nullptr,
                           const int32_t *dev_frames = nullptr);

// K4-P: offsets[f] = exclusive scan of frame_bytes (offsets[nframes] = total) and the frames
// copied back to back into packed[] -- the stream order flake_encode_frame's callers write.
hipError_t launch_pack_frames(hipStream_t st, const uint8_t *frames, int64_t frame_stride,
                              const int32_t *frame_bytes, int nframes, long long *offsets,
                              uint8_t *packed);

// Ragged (VBS) batches on the device (k4_assemble.hip).  A piece is k eighths of its block: eight
// bins of equal piece length, bin k-1 with a fixed range of frame slots for the most pieces it
// can get (floor(8 / k) per block).  Host constants per bin:
struct VbsBins {
    int n[8];                 // piece length k * block_size / 8
    int cap[8];               // frame slots: floor(8 / k) * nblocks
    int slot0[8];             // first frame slot (subframe-indexed workspaces: slot0 * channels)
    long long smp_off[8];     // first sample of the bin's rows in smp[] (int32 units)
    long long stride[8];      // fhip_frame_stride of the piece length
    long long fr_off[8];      // byte offset of the bin's frames in frames[]
    long long slot[8];        // residual-section slot bytes of the piece length
    long long bits_off[8];    // byte offset of the bin's sections in rice_bits[]
};
// One launch over ALL bins for the kernels whose code does not depend on the piece length (K1's
// wave-typed kernel, K2, K4): the chain walk of K1, the Levinson recursion of K2 and the serial
// header / CRC phases of K4 are latency, not throughput, and eight launches of a few hundred
// frames each paid it eight times.  Workgroup -> (bin, unit inside the bin) through wg0[]; the
// unit-indexed arrays (info, autoc, coefs ...) are the handle's whole workspaces, indexed by
// unit0[bin] + local unit; sample rows start at smp_off[bin].
struct MultiBin {
    int nbins;                // 0: a plain launch
    const int32_t *cnt;       // device: live units (subframes for K1 / K2, frames for K4) of entry k at
    int cnt_ix[8];            //   cnt[cnt_ix[k]]: entries may list the bins in any order (K1: longest first)
    int wg0[9];               // first workgroup of bin k; wg0[nbins] = the grid
    int n[8];
    int unit0[8];             // K1 / K2: first subframe of the bin; K4: first frame slot
    int cap[8];               // units the bin can hold
    int narrow[8];            // K1: the bin's rows may be 16-bit (K0's records say which)
    long long smp_off[8];
    double c[8];              // K1: the window constant of lpc.c:34 for the bin's n
    long long stride[8], fr_off[8], slot[8], bits_off[8];     // K4
    int vsize[8];                                              // K4: verbatim size (encode.c:521-527)
};
// K0 over all bins of a stereo batch (unit0 = first frame slot of the bin, cnt = live frames, wg0 from
// prepare_bins_workgroups): frame_src[slot] = the piece's offset in pcm
bool prepare_bins_supported(const fhip_params &p, const int *n, int nbins);
int prepare_bins_workgroups(int n, int cap, int nmax);     // nmax: the batch's longest bin
hipError_t launch_prepare_bins(hipStream_t st, const fhip_params &p, const int32_t *pcm, const MultiBin &mb,
                               int32_t *smp, fhip_subframe_info *info, const long long *frame_src);
hipError_t launch_autocorr_bins(hipStream_t st, const MultiBin &mb, const int32_t *smp, int max_order,
                                double *autoc, const fhip_subframe_info *info,
                                const autocorr_lpc_out *lpc_out);
bool autocorr_bins_supported(int max_order, const int *n, int nbins);
hipError_t launch_lpc_bins(hipStream_t st, const MultiBin &mb, const double *autoc, int max_order,
                           int precision, int omethod, int32_t *coefs, int32_t *shift,
                           int32_t *opt_order, int32_t *fin);
hipError_t launch_assemble_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *pcm,
                                const fhip_subframe_info *info, const uint8_t *rice, uint8_t *frames,
                                int32_t *frame_bytes, const uint32_t *numbers, const long long *frame_src);

// CNT_*: layout of the device-side counts k_vbs_plan leaves (int32[24])
constexpr int VBS_CNT_FRAMES = 0, VBS_CNT_SUB = 8, VBS_CNT_ALL = 16, VBS_CNT_WORDS = 24;
hipError_t launch_vbs_plan(hipStream_t st, const int32_t *nfr, const int32_t *sizes, int nblocks,
                           int block_size, int nch, uint32_t first_number, const VbsBins &bins,
                           int32_t *cnt, int32_t *order, long long *frame_src, long long *src_off,
                           uint32_t *numbers, int32_t *first);
// The frames of all bins packed in stream order: order[i] = slot of the stream's i-th frame,
// src_off[slot] = byte offset of that slot's frame in frames[] (4-byte aligned), *dev_frames of them
// (<= max_frames); offsets[i] / offsets[count] as in launch_pack_frames; stream_bytes[i] (optional)
// = size of the stream's i-th frame; totals[4] = {frames, bytes, largest frame, stream > cap}.
hipError_t launch_pack_frames_perm(hipStream_t st, const uint8_t *frames, const long long *src_off,
                                   const int32_t *frame_bytes, const int32_t *order, int max_frames,
                                   const int32_t *dev_frames, long long *offsets, uint8_t *packed,
                                   long long cap, int32_t *stream_bytes, long long *totals);
hipError_t launch_vbs_block_bytes(hipStream_t st, const int32_t *first, const long long *offsets, int nblocks,
                                  int32_t *block_bytes, int32_t *block_frames);

// K-vbs: split_frame_v1 (vbs.c:36-83) for nblocks blocks: nframes_out [nblocks],
// sizes_out [nblocks][8].
hipError_t launch_vbs_split(hipStream_t st, const int32_t *pcm, int nblocks, int block_size,
                            int nch, int32_t *nframes_out, int32_t *sizes_out);

// Dynamic-LDS need of K3 for a block size (0 if unsupported).
size_t encode_lds_bytes(int n);

// Fast-path geometry of K3 for a block size: C samples per thread, T threads, n = C*T.
bool fast_geometry(const fhip_params &p, int n, int *C, int *T);

}  // namespace fhip
