// api.hip -- the C ABI of libflakehip.so (include/flakehip.h) on top of the
// gfx950 kernels.  No allocation happens after fhip_create(); every entry
// point maps HIP failures to FHIP_E_HIP and never throws.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "flakehip.h"
#include "kernels.h"

struct fhip_ctx {
    int device = 0;
    fhip_params p{};
    int max_frames = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // device workspaces, sized for max_frames * channels subframes of p.block_size
    int32_t *d_smp = nullptr;         // [nsub][n]
    double  *d_autoc = nullptr;       // [nsub][33]
    int32_t *d_coefs = nullptr;       // [nsub][32][32]
    int32_t *d_shift = nullptr;       // [nsub][32]
    int32_t *d_opt = nullptr;         // [nsub]
    int32_t *d_fin = nullptr;         // [nsub][FIN_STRIDE]
    int32_t *d_tilectr = nullptr;     // [nsub / 32 + 1] K1's per-tile arrival counters (lag-split launches), zeroed once
    fhip_subframe_info *d_k0rec = nullptr;   // [nsub] K0's records (obits, wasted, ch_mode, row flag): K1 and
                                             // K3 read them here, K3 copies them into the caller's info[]
    // staging for the host-pointer entry points
    int32_t *d_pcm = nullptr;
    fhip_subframe_info *d_info = nullptr;
    int32_t *d_res = nullptr;
    uint8_t *d_bits = nullptr;
    size_t d_bits_bytes = 0;
    uint8_t *d_frames = nullptr;
    size_t d_frames_bytes = 0;
    int32_t *d_fbytes = nullptr;
    uint32_t *d_fnum = nullptr;
    uint8_t *d_packed = nullptr;      // fhip_encode_frames_packed: the frames back to back
    size_t d_packed_bytes = 0;
    long long *d_offsets = nullptr;   // [max_frames + 1]
    long long packed_ready = 0;       // bytes waiting in d_packed between _begin and _fetch
    const int32_t *uploaded_pcm = nullptr;   // fhip_frames_packed_upload: this host batch already lies in d_pcm
    size_t uploaded_vals = 0;
    hipEvent_t ev_fetch = nullptr;           // fhip_frames_packed_fetch_async: behind the download of d_packed (on aux[0])
    bool fetch_pending = false;
    // variable-block-size batches (fhip_encode_blocks_vbs_dev): the piece tables k_vbs_plan leaves
    size_t ws_frames = 0;             // frame capacity of the subframe-indexed workspaces (>= max_frames:
                                      // a VBS handle has 20 slots per block, eight bins of fixed capacity)
    long long *d_srcoff = nullptr, *d_frame_src = nullptr, *d_totals = nullptr;
    int32_t *d_order = nullptr, *d_vcnt = nullptr, *d_first = nullptr;
    int32_t *d_stream_bytes = nullptr, *d_blk_bytes = nullptr, *d_blk_frames = nullptr;

    // two internal streams for the split-batch overlap (run_pipeline)
    static constexpr int NAUX = 4;       // (run_pipeline's split uses the first two, the VBS groups all)
    hipStream_t aux[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {nullptr, nullptr, nullptr, nullptr};
    bool overlap = false;     // measured slower on C2 (K1 is chain-length bound): opt-in, FHIP_OVERLAP=1

    // fhip_prepare_ahead: K0 of the NEXT batch on its own stream while this batch's K1 runs.
    // Two sample buffers and two sets of K0 records of their own (allocated at first use), so
    // that nothing an in-flight batch reads is written: ahead[b] is written by K0 on `pre`
    // (after ev_enc[b]: the K3 that last read it) and read by K1/K3 on `stream` (after ev_prep[b]).
    hipStream_t pre = nullptr;
    int32_t *d_smp_ahead[2] = {nullptr, nullptr};
    fhip_subframe_info *d_prep[2] = {nullptr, nullptr};
    hipEvent_t ev_prep[2] = {nullptr, nullptr}, ev_enc[2] = {nullptr, nullptr};
    bool enc_recorded[2] = {false, false};
    int ahead_next = 0;
    hipEvent_t ev_k1 = nullptr;      // after K1 of the batch queued last (experiment: gate K0 on it)
    bool k1_recorded = false;
    int ahead_gate = 0;              // FHIP_AHEAD_GATE: 0 none, 1 the next K0 starts when K1 has ended
    struct Ahead { bool valid = false; const int32_t *pcm = nullptr; int nframes = 0, n = 0, buf = 0;
                   bool narrow = false; } ahead;

    bool profiling = false;
    struct KTime { const char *name; double ms = 0; int launches = 0; };
    std::vector<KTime> ktimes;
    struct Pending { int idx; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;

    std::string err;
};

namespace {

const char *const kKernelNames[6] = {"k_prepare", "k_autocorr", "k_lpc", "k_encode", "k_assemble",
                                     "k_order_search"};

int fail_hip(fhip_ctx *c, hipError_t e, const char *what)
{
    if (c) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
    }
    (void)hipGetLastError();
    return FHIP_E_HIP;
}

int fail(fhip_ctx *c, int code, const char *what)
{
    if (c) c->err = what;
    return code;
}

#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call); \
    } while (0)

// flake_validate_params(), encode.c:268-373, on the fields we mirror.
int validate(const fhip_params *p)
{
    if (!p) return -1;
    if (p->channels < 1 || p->channels > FHIP_MAX_CH) return -1;
    if (p->sample_rate < 1 || p->sample_rate > 655350) return -1;
    if (p->bits_per_sample < 4 || p->bits_per_sample > 32) return -1;
    if (p->order_method < 0 || p->order_method > 6) return -1;
    if (p->stereo_method < 0 || p->stereo_method > 1) return -1;
    if (p->block_size < 16 || p->block_size > 65535) return -1;
    if (p->prediction_type < 0 || p->prediction_type > 2) return -1;
    if (p->min_prediction_order > p->max_prediction_order) return -1;
    if (p->prediction_type == 1) {
        if (p->min_prediction_order < 0 || p->min_prediction_order > 4) return -1;
        if (p->max_prediction_order < 0 || p->max_prediction_order > 4) return -1;
    } else {
        if (p->min_prediction_order < 1 || p->min_prediction_order > 32) return -1;
        if (p->max_prediction_order < 1 || p->max_prediction_order > 32) return -1;
    }
    if (p->min_partition_order > p->max_partition_order) return -1;
    if (p->min_partition_order < 0 || p->min_partition_order > 8) return -1;
    if (p->max_partition_order < 0 || p->max_partition_order > 8) return -1;
    if (p->variable_block_size < 0 || p->variable_block_size > 1) return -1;
    if (p->variable_block_size > 0 && !p->allow_vbs) return -1;
    if (p->block_size < 128 && p->allow_vbs) return -1;
    if (p->lpc_precision < 2 || p->lpc_precision > 15) return -1;
    return 0;
}

struct Prof {
    fhip_ctx *c;
    int idx;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    Prof(fhip_ctx *ctx, int i, hipStream_t on = nullptr) : c(ctx), idx(i), st(on ? on : ctx->stream)
    {
        if (!c->profiling) return;
        a = take();
        b = take();
        if (a && b) (void)hipEventRecord(a, st);
    }
    hipEvent_t take()
    {
        if (!c->event_pool.empty()) {
            hipEvent_t e = c->event_pool.back();
            c->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    ~Prof()
    {
        if (!c->profiling || !a || !b) return;
        (void)hipEventRecord(b, st);
        c->pending.push_back({idx, a, b});
    }
};

void drain_profile(fhip_ctx *c)
{
    std::vector<fhip_ctx::Pending> later;
    for (auto &pd : c->pending) {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, pd.a, pd.b);
        if (e == hipErrorNotReady) {              // a prepare-ahead launch still in flight
            (void)hipGetLastError();
            later.push_back(pd);
            continue;
        }
        if (e == hipSuccess) {
            c->ktimes[pd.idx].ms += ms;
            c->ktimes[pd.idx].launches += 1;
        } else {
            (void)hipGetLastError();
        }
        c->event_pool.push_back(pd.a);
        c->event_pool.push_back(pd.b);
    }
    c->pending.swap(later);
}

// The four launches of one range of frames.  All pointers are device pointers
// and already offset to the range; `prof` brackets each launch with events.
struct FrameOut { uint8_t *frames; int64_t stride; int32_t *bytes; uint32_t first; const uint32_t *numbers = nullptr; };
// One bin of a variable-block-size batch: its pieces lie where the caller's blocks lie (frame_src),
// their number is on the device (dev_frames / dev_sub); nframes is then the bin's capacity and
// hint_frames the count the host-side choices (which K1, 16-bit rows) are made for.
struct Ragged { const long long *frame_src; const int32_t *dev_frames, *dev_sub; int hint_frames; };

static int run_range(fhip_ctx *c, hipStream_t st, bool prof, const int32_t *pcm, int nframes, int n,
                     fhip_subframe_info *info, int32_t *residual, uint8_t *bits,
                     int64_t slot_bytes, int32_t *smp, double *autoc, size_t sub0,
                     const FrameOut &fo, bool want_rows = false,
                     const fhip_subframe_info *prepared = nullptr, bool prepared_narrow = false,
                     const Ragged *rg = nullptr)
{
    const fhip_params &p = c->p;
    const int nsub = nframes * p.channels;
    const int nsub_hint = rg ? rg->hint_frames * p.channels : nsub;
    const long long *frame_src = rg ? rg->frame_src : nullptr;
    const int32_t *dev_frames = rg ? rg->dev_frames : nullptr, *dev_sub = rg ? rg->dev_sub : nullptr;
    const bool lpc_path = (p.prediction_type == 2) && (n > p.max_prediction_order) && n >= 5;
    int32_t *coefs = c->d_coefs + sub0 * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
    int32_t *shift = c->d_shift + sub0 * FHIP_MAX_ORDER;
    int32_t *opt = c->d_opt + sub0;
    int32_t *fin = c->d_fin + sub0 * fhip::FIN_STRIDE;
    struct MaybeProf {
        fhip_ctx *c; bool on; Prof *p;
        MaybeProf(fhip_ctx *cc, bool o, int i) : c(cc), on(o), p(o ? new Prof(cc, i) : nullptr) {}
        ~MaybeProf() { delete p; }
    };
    // stereo batches in whole tiles: K0 only decides (ch_mode, wasted bits), the K1
    // producers apply that to the PCM they stream anyway and write smp
    // `prepared`: K0 of this batch already ran (fhip_prepare_ahead) into smp and these records
    const bool fused = !prepared && !rg && lpc_path && fhip::autocorr_fuses_prepare(p, nsub, n);
    // rows of 16-bit samples where every kernel of this batch reads them that way and
    // nobody outside asked for the int32 rows
    const bool narrow = prepared ? prepared_narrow
                                 : (!fused && !want_rows && fhip::narrow_rows_ok(p, nsub_hint, n, lpc_path));
    // K0's records never live in the caller's info[] (K3's pointers to the two do not alias)
    fhip_subframe_info *own_rec = c->d_k0rec + sub0;
    const fhip_subframe_info *k0rec = prepared ? prepared : own_rec;
    if (!prepared) {
        MaybeProf pr(c, prof, 0);
        HIP_TRY(c, fhip::launch_prepare(st, p, pcm, nframes, n, smp, own_rec, fused, narrow, frame_src, dev_frames));
    }
    if (lpc_path) {
        // K2 rides on K1's tail where K1 is the wave-typed kernel and the order fits registers
        // (the arrival counters of a lag-split launch are indexed by the tile within the launch: a range of frames
        // that does not start at the handle's first subframe -- the split-batch overlap -- keeps its own K2 launch)
        const bool ctr_ok = sub0 == 0 && !rg;
        const bool lpc_tail = !fused && fhip::autocorr_does_lpc(nsub_hint, n, p.max_prediction_order, ctr_ok);
        const fhip::autocorr_lpc_out lo{p.lpc_precision, p.order_method, coefs, shift, opt, fin, ctr_ok ? c->d_tilectr : nullptr};
        {
            MaybeProf pr(c, prof, 1);
            HIP_TRY(c, fhip::launch_autocorr(st, smp, nsub, n, p.max_prediction_order, autoc,
                                             fused ? pcm : nullptr, fused ? smp : nullptr, k0rec,
                                             lpc_tail ? &lo : nullptr, narrow, dev_sub, nsub_hint));
        }
        if (!lpc_tail) {
            MaybeProf pr(c, prof, 2);
            HIP_TRY(c, fhip::launch_lpc(st, autoc, nsub, p.max_prediction_order, p.lpc_precision,
                                        p.order_method, coefs, shift, opt, fin, dev_sub));
        }
    }
    if (c->ev_k1 && st == c->stream) {
        HIP_TRY(c, hipEventRecord(c->ev_k1, st));
        c->k1_recorded = true;
    }
    // order searches: the table of candidates and the method's walk over it in a kernel of
    // their own; K3 then encodes the one chosen row like a MAX / EST row
    const bool searched = lpc_path && fhip::order_search_supported(p, n);
    if (searched) {
        MaybeProf pr(c, prof, 5);
        HIP_TRY(c, fhip::launch_order_search(st, p, smp, nsub, n, coefs, shift, opt, fin, k0rec, narrow, dev_sub));
    }
    {
        MaybeProf pr(c, prof, 3);
        HIP_TRY(c, fhip::launch_encode(st, p, smp, nsub, n, coefs, shift, opt, fin, info, residual,
                                       bits, slot_bytes, -1, 0, narrow, k0rec, searched, dev_sub));
    }
    if (fo.frames) {
        MaybeProf pr(c, prof, 4);
        const uint32_t step = p.allow_vbs ? (uint32_t)n : 1u;
        HIP_TRY(c, fhip::launch_assemble(st, p, pcm, nframes, n, info, bits, slot_bytes, fo.frames,
                                         fo.stride, fo.bytes, fo.first, step, fo.numbers, frame_src, dev_frames));
    }
    return FHIP_OK;
}

// One batch.  Optionally (FHIP_OVERLAP=1) a large batch is cut into two frame
// ranges that run on two internal streams forked from (and joined back into)
// the caller's stream.  Measured on configs[1]: 0.342 ms/step split vs 0.267
// ms/step in order -- K1's time is the length of one chain walk, not the
// number of subframes, so two half-size K1 launches cost two walks.  Off by
// default.
int run_pipeline(fhip_ctx *c, const int32_t *pcm, int nframes, int n,
                 fhip_subframe_info *info, int32_t *residual, uint8_t *bits,
                 int64_t slot_bytes, int32_t *samples_out, double *autoc_out,
                 const FrameOut &fo = FrameOut{nullptr, 0, nullptr, 0}, bool want_rows = false)
{
    want_rows = want_rows || samples_out != nullptr;     // somebody reads FlacSubframe.samples as int32
    const fhip_params &p = c->p;
    if (c->ahead.valid) {
        // whatever was prepared ahead is ordered before this batch either way
        const fhip_ctx::Ahead a = c->ahead;
        c->ahead.valid = false;
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_prep[a.buf], 0));
        if (a.pcm == pcm && a.nframes == nframes && a.n == n && !want_rows && !autoc_out) {
            int rc = run_range(c, c->stream, c->profiling, pcm, nframes, n, info, residual, bits,
                               slot_bytes, c->d_smp_ahead[a.buf], c->d_autoc, 0, fo, false,
                               c->d_prep[a.buf], a.narrow);
            if (rc != FHIP_OK) return rc;
            HIP_TRY(c, hipEventRecord(c->ev_enc[a.buf], c->stream));
            c->enc_recorded[a.buf] = true;
            return FHIP_OK;
        }
    }
    int32_t *smp = samples_out ? samples_out : c->d_smp;
    double *autoc = autoc_out ? autoc_out : c->d_autoc;
    const size_t nch = (size_t)p.channels;
    const bool split = !c->profiling && c->overlap && nframes >= 512 && c->aux[0] && c->aux[1];
    if (!split)
        return run_range(c, c->stream, c->profiling, pcm, nframes, n, info, residual, bits,
                         slot_bytes, smp, autoc, 0, fo, want_rows);

    HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
    const int parts[3] = {0, nframes / 2, nframes};
    for (int h = 0; h < 2; h++) {
        HIP_TRY(c, hipStreamWaitEvent(c->aux[h], c->ev_fork, 0));
        const size_t f0 = (size_t)parts[h], nf = (size_t)(parts[h + 1] - parts[h]);
        const size_t sub0 = f0 * nch;
        int rc = run_range(c, c->aux[h], false, pcm + f0 * n * nch, (int)nf, n, info + sub0,
                           residual ? residual + sub0 * n : nullptr,
                           bits ? bits + sub0 * (size_t)slot_bytes : nullptr, slot_bytes,
                           smp + sub0 * n, autoc + sub0 * FHIP_MAX_LAGS, sub0,
                           FrameOut{fo.frames ? fo.frames + f0 * (size_t)fo.stride : nullptr, fo.stride,
                                    fo.bytes ? fo.bytes + f0 : nullptr,
                                    fo.first + (uint32_t)f0 * (p.allow_vbs ? (uint32_t)n : 1u),
                                    fo.numbers ? fo.numbers + f0 : nullptr}, want_rows);
        if (rc != FHIP_OK) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_join[h], c->aux[h]));
    }
    for (int h = 0; h < 2; h++) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join[h], 0));
    return FHIP_OK;
}

int check_batch(fhip_ctx *c, const fhip_batch *b, bool host = false)
{
    // host entry: frames alone are a complete result, info / rice_bits optional then
    const bool lean = host && b && b->frames;
    if (!c || !b || !b->pcm || (!b->info && !lean)) return fail(c, FHIP_E_INVALID, "null batch argument");
    if (b->nframes < 0 || b->nframes > c->max_frames)
        return fail(c, FHIP_E_INVALID, "nframes exceeds the handle's max_frames");
    if (b->block_size < 1 || b->block_size > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "block_size out of range (encode.c:987)");
    if (b->block_size > FHIP_MAX_BLOCK)
        return fail(c, FHIP_E_UNSUPPORTED, "block_size above FHIP_MAX_BLOCK");
    if ((b->rice_bits || lean) && (b->rice_slot_bytes < 4 || (b->rice_slot_bytes & 3)))
        return fail(c, FHIP_E_INVALID, "rice_slot_bytes must be a positive multiple of 4");
    if (b->frames) {
        if ((!b->rice_bits && !lean) || !b->frame_bytes)
            return fail(c, FHIP_E_INVALID, "frames need rice_bits and frame_bytes");
        if ((b->frame_stride & 3) || b->frame_stride < fhip_frame_stride(&c->p, b->block_size))
            return fail(c, FHIP_E_INVALID, "frame_stride too small or not a multiple of 4");
    }
    return FHIP_OK;
}

}  // namespace

extern "C" {

int64_t fhip_frame_stride(const fhip_params *p, int block_size)
{
    if (!p || block_size < 1) return 0;
    const int64_t bps = p->bits_per_sample, n = block_size;
    const int64_t v = (p->channels == 2) ? 16 + ((n * (bps + bps + 1) + 7) >> 3)
                                         : 16 + ((n * p->channels * bps + 7) >> 3);
    return (v + 8 + 3) & ~(int64_t)3;
}

int fhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return FHIP_E_HIP;
    }
    return n;
}

const char *fhip_version(void) { return "flake-amd 0.1 (gfx950)"; }

const char *fhip_strerror(int code)
{
    switch (code) {
    case FHIP_OK: return "ok";
    case FHIP_E_GENERIC: return "generic failure";
    case FHIP_E_HIP: return "HIP runtime error";
    case FHIP_E_UNSUPPORTED: return "parameters not supported by the HIP layer";
    case FHIP_E_INVALID: return "invalid parameters";
    case FHIP_E_NOMEM: return "out of memory";
    default: return code > 0 ? "ok" : "unknown error";
    }
}

const char *fhip_last_error(const fhip_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int fhip_create(fhip_ctx **out, int device, const fhip_params *p, int max_frames)
{
    if (!out) return FHIP_E_INVALID;
    *out = nullptr;
    if (validate(p) < 0 || max_frames < 1) return FHIP_E_INVALID;
    if (p->block_size > FHIP_MAX_BLOCK) return FHIP_E_UNSUPPORTED;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return FHIP_E_HIP;
    }
    fhip_ctx *c = new (std::nothrow) fhip_ctx();
    if (!c) return FHIP_E_NOMEM;
    c->device = device;
    c->p = *p;
    c->max_frames = max_frames;
    for (int i = 0; i < 6; i++) c->ktimes.push_back({kKernelNames[i], 0.0, 0});

    // subframe-indexed workspaces: a variable-block-size handle keeps eight bins of fixed capacity
    // (pieces of k eighths: floor(8 / k) per block, 20 slots per block in all); the sample-indexed
    // ones need no more than before (a bin's pieces never exceed its blocks' samples)
    c->ws_frames = p->variable_block_size ? (size_t)20 * (((size_t)max_frames + 7) / 8) : (size_t)max_frames;
    if (c->ws_frames < (size_t)max_frames) c->ws_frames = (size_t)max_frames;
    const size_t nsub = c->ws_frames * p->channels;
    const size_t nsmp = (size_t)max_frames * p->channels;
    const size_t n = (size_t)p->block_size;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    for (int h = 0; h < fhip_ctx::NAUX && e == hipSuccess; h++) {
        e = hipStreamCreateWithFlags(&c->aux[h], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[h], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (const char *v = getenv("FHIP_OVERLAP")) c->overlap = (v[0] == '1');
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_smp, nsmp * n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_autoc, nsub * FHIP_MAX_LAGS * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_coefs, nsub * FHIP_MAX_ORDER * FHIP_MAX_ORDER * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_shift, nsub * FHIP_MAX_ORDER * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_opt, nsub * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_fin, nsub * fhip::FIN_STRIDE * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_tilectr, (nsub / 32 + 2) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(c->d_tilectr, 0, (nsub / 32 + 2) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_k0rec, nsub * sizeof(fhip_subframe_info));
    if (e == hipSuccess) e = hipMemset(c->d_k0rec, 0, nsub * sizeof(fhip_subframe_info));
    if (e == hipSuccess) e = hipMemset(c->d_fin, 0, nsub * fhip::FIN_STRIDE * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(c->d_coefs, 0, nsub * FHIP_MAX_ORDER * FHIP_MAX_ORDER * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(c->d_shift, 0, nsub * FHIP_MAX_ORDER * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(c->d_opt, 0, nsub * sizeof(int32_t));
    // (the fills run on the null stream; the handle's streams are non-blocking: wait for them)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        int rc = (e == hipErrorOutOfMemory) ? FHIP_E_NOMEM : FHIP_E_HIP;
        fhip_destroy(c);
        (void)hipGetLastError();
        return rc;
    }
    c->stream = c->own_stream;
    *out = c;
    return FHIP_OK;
}

void fhip_destroy(fhip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->pre) (void)hipStreamSynchronize(c->pre);
    drain_profile(c);
    for (hipEvent_t ev : c->event_pool) (void)hipEventDestroy(ev);
    void *bufs[] = {c->d_smp, c->d_autoc, c->d_coefs, c->d_shift, c->d_opt, c->d_fin, c->d_k0rec, c->d_tilectr,
                    c->d_pcm, c->d_info, c->d_res, c->d_bits, c->d_frames, c->d_fbytes, c->d_fnum,
                    c->d_packed, c->d_offsets, c->d_srcoff, c->d_frame_src, c->d_totals, c->d_order,
                    c->d_vcnt, c->d_first, c->d_stream_bytes, c->d_blk_bytes, c->d_blk_frames};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (int h = 0; h < fhip_ctx::NAUX; h++) {
        if (c->aux[h]) { (void)hipStreamSynchronize(c->aux[h]); (void)hipStreamDestroy(c->aux[h]); }
        if (c->ev_join[h]) (void)hipEventDestroy(c->ev_join[h]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->pre) { (void)hipStreamSynchronize(c->pre); (void)hipStreamDestroy(c->pre); }
    if (c->ev_k1) (void)hipEventDestroy(c->ev_k1);
    if (c->ev_fetch) { (void)hipEventSynchronize(c->ev_fetch); (void)hipEventDestroy(c->ev_fetch); }
    for (int h = 0; h < 2; h++) {
        if (c->ev_prep[h]) (void)hipEventDestroy(c->ev_prep[h]);
        if (c->ev_enc[h]) (void)hipEventDestroy(c->ev_enc[h]);
        if (c->d_smp_ahead[h]) (void)hipFree(c->d_smp_ahead[h]);
        if (c->d_prep[h]) (void)hipFree(c->d_prep[h]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int fhip_set_stream(fhip_ctx *c, void *hip_stream)
{
    if (!c) return FHIP_E_INVALID;
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return FHIP_OK;
}

int fhip_sync(fhip_ctx *c)
{
    if (!c) return FHIP_E_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_profile(c);
    return FHIP_OK;
}

int fhip_set_profiling(fhip_ctx *c, int on)
{
    if (!c) return FHIP_E_INVALID;
    c->profiling = on != 0;
    return FHIP_OK;
}

int fhip_get_kernel_times(fhip_ctx *c, const char **names, double *ms, int *launches,
                          int cap, int reset)
{
    if (!c) return FHIP_E_INVALID;
    int k = 0;
    for (auto &t : c->ktimes) {
        if (k < cap) {
            if (names) names[k] = t.name;
            if (ms) ms[k] = t.ms;
            if (launches) launches[k] = t.launches;
        }
        k++;
        if (reset) { t.ms = 0; t.launches = 0; }
    }
    return k;
}

int fhip_encode_subframes_dev(fhip_ctx *c, const fhip_batch *b)
{
    int rc = check_batch(c, b);
    if (rc != FHIP_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    return run_pipeline(c, b->pcm, b->nframes, b->block_size, b->info, b->residual,
                        b->rice_bits, b->rice_slot_bytes, b->samples, b->autoc,
                        FrameOut{b->frames, b->frame_stride, b->frame_bytes, b->first_frame_number,
                                 b->frame_numbers});
}

int fhip_prepare_ahead(fhip_ctx *c, const fhip_batch *b)
{
    if (!c || !b || !b->pcm) return fail(c, FHIP_E_INVALID, "null batch argument");
    if (b->nframes < 0 || b->nframes > c->max_frames)
        return fail(c, FHIP_E_INVALID, "nframes exceeds the handle's max_frames");
    if (b->block_size < 1 || b->block_size > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "block_size out of range (encode.c:987)");
    if (b->nframes == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const fhip_params &p = c->p;
    const size_t cap = (size_t)c->max_frames * p.channels;
    if (!c->pre) {
        // FHIP_AHEAD_PRIO=1 (measurements): the feeder's stream at the lowest priority, so that K1's one workgroup per
        // CU is placed before K0's workgroups take the CU's registers (profiles/r04_overlap_*.txt)
        int lo = 0, hi = 0;
        const char *pv = getenv("FHIP_AHEAD_PRIO");
        if (pv && pv[0] == '1' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            HIP_TRY(c, hipStreamCreateWithPriority(&c->pre, hipStreamNonBlocking, lo));
        else
            HIP_TRY(c, hipStreamCreateWithFlags(&c->pre, hipStreamNonBlocking));
        if (const char *v = getenv("FHIP_AHEAD_GATE")) c->ahead_gate = atoi(v);
        if (c->ahead_gate) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_k1, hipEventDisableTiming));
        for (int h = 0; h < 2; h++) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_prep[h], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_enc[h], hipEventDisableTiming));
            HIP_TRY(c, hipMalloc((void **)&c->d_smp_ahead[h], cap * (size_t)p.block_size * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc((void **)&c->d_prep[h], cap * sizeof(fhip_subframe_info)));
            HIP_TRY(c, hipMemset(c->d_prep[h], 0, cap * sizeof(fhip_subframe_info)));
        }
        // hipMemset runs on the null stream, which does not order with this handle's
        // non-blocking streams: finish the fills before any kernel may write these buffers
        HIP_TRY(c, hipDeviceSynchronize());
    }
    if (c->ahead.valid) {          // a second hint without an encode in between: the first one is dropped
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_prep[c->ahead.buf], 0));
        c->ahead.valid = false;
    }
    const int buf = c->ahead_next;
    c->ahead_next ^= 1;
    const int n = b->block_size, nsub = b->nframes * p.channels;
    const bool lpc_path = (p.prediction_type == 2) && (n > p.max_prediction_order) && n >= 5;
    const bool narrow = fhip::narrow_rows_ok(p, nsub, n, lpc_path);
    if (c->enc_recorded[buf]) HIP_TRY(c, hipStreamWaitEvent(c->pre, c->ev_enc[buf], 0));
    if (c->ahead_gate == 1 && c->k1_recorded) HIP_TRY(c, hipStreamWaitEvent(c->pre, c->ev_k1, 0));
    {
        Prof pr(c, 0, c->pre);
        HIP_TRY(c, fhip::launch_prepare(c->pre, p, b->pcm, b->nframes, n, c->d_smp_ahead[buf],
                                        c->d_prep[buf], false, narrow));
    }
    HIP_TRY(c, hipEventRecord(c->ev_prep[buf], c->pre));
    c->ahead.valid = true;
    c->ahead.pcm = b->pcm;
    c->ahead.nframes = b->nframes;
    c->ahead.n = n;
    c->ahead.buf = buf;
    c->ahead.narrow = narrow;
    return FHIP_OK;
}

// Lazily sized staging buffers for the host-pointer entry points.
static int ensure_staging(fhip_ctx *c, size_t bits_bytes)
{
    const size_t nsub = (size_t)c->max_frames * c->p.channels;
    const size_t n = (size_t)c->p.block_size;
    if (!c->d_pcm) HIP_TRY(c, hipMalloc((void **)&c->d_pcm, nsub * n * sizeof(int32_t)));
    if (!c->d_info) HIP_TRY(c, hipMalloc((void **)&c->d_info, c->ws_frames * c->p.channels * sizeof(fhip_subframe_info)));
    if (!c->d_res) HIP_TRY(c, hipMalloc((void **)&c->d_res, nsub * n * sizeof(int32_t)));
    if (bits_bytes > c->d_bits_bytes) {
        if (c->d_bits) (void)hipFree(c->d_bits);
        c->d_bits = nullptr;
        c->d_bits_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_bits, bits_bytes));
        c->d_bits_bytes = bits_bytes;
    }
    return FHIP_OK;
}

int fhip_encode_subframes(fhip_ctx *c, const fhip_batch *b)
{
    int rc = check_batch(c, b, true);
    if (rc != FHIP_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nch = (size_t)c->p.channels, n = (size_t)b->block_size;
    const size_t nsub = (size_t)b->nframes * nch;
    const bool want_bits = b->rice_bits || b->frames;      // K4 reads the sections on the device
    const size_t bits_bytes = want_bits ? nsub * (size_t)b->rice_slot_bytes : 0;
    rc = ensure_staging(c, bits_bytes);
    if (rc != FHIP_OK) return rc;
    if (nsub == 0) return FHIP_OK;

    double *d_autoc_out = b->autoc ? c->d_autoc : nullptr;
    // a section that does not fit its slot leaves the slot untouched: what comes back for it is zeros
    if (b->rice_bits) HIP_TRY(c, hipMemsetAsync(c->d_bits, 0, bits_bytes, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_pcm, b->pcm, nsub * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, nsub * sizeof(fhip_subframe_info), c->stream));
    FrameOut fo{nullptr, 0, nullptr, 0};
    if (b->frames) {
        const size_t fb = (size_t)b->nframes * (size_t)b->frame_stride;
        if (fb > c->d_frames_bytes) {
            if (c->d_frames) (void)hipFree(c->d_frames);
            c->d_frames = nullptr; c->d_frames_bytes = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_frames, fb));
            c->d_frames_bytes = fb;
        }
        if (!c->d_fbytes) HIP_TRY(c, hipMalloc((void **)&c->d_fbytes, c->ws_frames * sizeof(int32_t)));
        fo = FrameOut{c->d_frames, b->frame_stride, c->d_fbytes, b->first_frame_number, nullptr};
        if (b->frame_numbers) {
            if (!c->d_fnum) HIP_TRY(c, hipMalloc((void **)&c->d_fnum, c->ws_frames * sizeof(uint32_t)));
            HIP_TRY(c, hipMemcpyAsync(c->d_fnum, b->frame_numbers, (size_t)b->nframes * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            fo.numbers = c->d_fnum;
        }
    }
    rc = run_pipeline(c, c->d_pcm, b->nframes, b->block_size, c->d_info,
                      b->residual ? c->d_res : nullptr, want_bits ? c->d_bits : nullptr,
                      b->rice_slot_bytes, nullptr, d_autoc_out, fo, b->samples != nullptr);
    if (rc != FHIP_OK) return rc;
    if (b->info)
        HIP_TRY(c, hipMemcpyAsync(b->info, c->d_info, nsub * sizeof(fhip_subframe_info), hipMemcpyDeviceToHost, c->stream));
    if (b->residual)
        HIP_TRY(c, hipMemcpyAsync(b->residual, c->d_res, nsub * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (b->rice_bits)
        HIP_TRY(c, hipMemcpyAsync(b->rice_bits, c->d_bits, bits_bytes, hipMemcpyDeviceToHost, c->stream));
    if (b->samples)
        HIP_TRY(c, hipMemcpyAsync(b->samples, c->d_smp, nsub * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (b->frames) {
        HIP_TRY(c, hipMemcpyAsync(b->frames, c->d_frames, (size_t)b->nframes * (size_t)b->frame_stride, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(b->frame_bytes, c->d_fbytes, (size_t)b->nframes * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    if (b->autoc)
        HIP_TRY(c, hipMemcpyAsync(b->autoc, c->d_autoc, nsub * FHIP_MAX_LAGS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return fhip_sync(c);
}

int fhip_frames_packed_upload(fhip_ctx *c, const fhip_batch *b)
{
    if (!c || !b || !b->pcm) return fail(c, FHIP_E_INVALID, "null argument");
    if (b->nframes < 0 || b->nframes > c->max_frames)
        return fail(c, FHIP_E_INVALID, "nframes exceeds the handle's max_frames");
    if (b->block_size < 1 || b->block_size > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "block_size out of range (encode.c:987)");
    c->uploaded_pcm = nullptr;
    if (b->nframes == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nvals = (size_t)b->nframes * (size_t)c->p.channels * (size_t)b->block_size;
    const int64_t stride = fhip_frame_stride(&c->p, b->block_size);
    int rc = ensure_staging(c, (size_t)b->nframes * c->p.channels * (size_t)((stride + 3) & ~(int64_t)3));
    if (rc != FHIP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_pcm, b->pcm, nvals * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->uploaded_pcm = b->pcm;
    c->uploaded_vals = nvals;
    return FHIP_OK;
}

int fhip_frames_packed_begin(fhip_ctx *c, const fhip_batch *b, int64_t *total_bytes)
{
    if (!c || !b || !b->pcm || !total_bytes || !b->frame_bytes)
        return fail(c, FHIP_E_INVALID, "null argument");
    if (b->nframes < 0 || b->nframes > c->max_frames)
        return fail(c, FHIP_E_INVALID, "nframes exceeds the handle's max_frames");
    if (b->block_size < 1 || b->block_size > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "block_size out of range (encode.c:987)");
    *total_bytes = 0;
    c->packed_ready = 0;
    if (b->nframes == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nch = (size_t)c->p.channels, n = (size_t)b->block_size;
    const size_t nsub = (size_t)b->nframes * nch;
    const int64_t stride = fhip_frame_stride(&c->p, b->block_size);
    const int64_t slot = (stride + 3) & ~(int64_t)3;            // a subframe's section fits the frame's slot
    int rc = ensure_staging(c, nsub * (size_t)slot);
    if (rc != FHIP_OK) return rc;
    const size_t fb = (size_t)b->nframes * (size_t)stride;
    if (fb > c->d_frames_bytes) {
        if (c->d_frames) (void)hipFree(c->d_frames);
        c->d_frames = nullptr; c->d_frames_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_frames, fb));
        c->d_frames_bytes = fb;
    }
    if (fb > c->d_packed_bytes) {
        if (c->d_packed) (void)hipFree(c->d_packed);
        c->d_packed = nullptr; c->d_packed_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_packed, fb));
        c->d_packed_bytes = fb;
    }
    if (!c->d_fbytes) HIP_TRY(c, hipMalloc((void **)&c->d_fbytes, c->ws_frames * sizeof(int32_t)));
    if (!c->d_offsets) HIP_TRY(c, hipMalloc((void **)&c->d_offsets, ((size_t)c->max_frames + 1) * sizeof(long long)));
    FrameOut fo{c->d_frames, stride, c->d_fbytes, b->first_frame_number, nullptr};
    if (b->frame_numbers) {
        if (!c->d_fnum) HIP_TRY(c, hipMalloc((void **)&c->d_fnum, c->ws_frames * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpyAsync(c->d_fnum, b->frame_numbers, (size_t)b->nframes * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        fo.numbers = c->d_fnum;
    }
    const bool uploaded = c->uploaded_pcm == b->pcm && c->uploaded_vals == nsub * n;
    c->uploaded_pcm = nullptr;
    if (c->fetch_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_fetch, 0));     // d_packed is still being read
    if (!uploaded)
        HIP_TRY(c, hipMemcpyAsync(c->d_pcm, b->pcm, nsub * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = run_pipeline(c, c->d_pcm, b->nframes, b->block_size, c->d_info, nullptr, c->d_bits, slot,
                      nullptr, nullptr, fo, false);
    if (rc != FHIP_OK) return rc;
    HIP_TRY(c, fhip::launch_pack_frames(c->stream, c->d_frames, stride, c->d_fbytes, b->nframes,
                                        c->d_offsets, c->d_packed));
    long long total = 0;
    HIP_TRY(c, hipMemcpyAsync(b->frame_bytes, c->d_fbytes, (size_t)b->nframes * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&total, c->d_offsets + b->nframes, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    if (b->info)
        HIP_TRY(c, hipMemcpyAsync(b->info, c->d_info, nsub * sizeof(fhip_subframe_info), hipMemcpyDeviceToHost, c->stream));
    rc = fhip_sync(c);
    if (rc != FHIP_OK) return rc;
    c->packed_ready = total;
    *total_bytes = total;
    return FHIP_OK;
}

int fhip_frames_packed_fetch(fhip_ctx *c, uint8_t *out, int64_t out_cap)
{
    if (!c || !out) return fail(c, FHIP_E_INVALID, "null argument");
    if (c->packed_ready > out_cap) return fail(c, FHIP_E_INVALID, "output buffer too small for the batch's frames");
    if (c->packed_ready > 0) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipMemcpy(out, c->d_packed, (size_t)c->packed_ready, hipMemcpyDeviceToHost));
    }
    c->packed_ready = 0;
    return FHIP_OK;
}

void *fhip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void fhip_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}

int fhip_host_register(void *p, size_t bytes)
{
    if (!p || bytes == 0) return FHIP_E_INVALID;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return FHIP_E_HIP;
    }
    return FHIP_OK;
}

int fhip_host_unregister(void *p)
{
    if (!p) return FHIP_E_INVALID;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return FHIP_E_HIP;
    }
    return FHIP_OK;
}

int fhip_frames_packed_fetch_async(fhip_ctx *c, uint8_t *out, int64_t out_cap)
{
    if (!c || !out) return fail(c, FHIP_E_INVALID, "null argument");
    if (c->packed_ready > out_cap) return fail(c, FHIP_E_INVALID, "output buffer too small for the batch's frames");
    if (c->packed_ready > 0) {
        HIP_TRY(c, hipSetDevice(c->device));
        if (!c->ev_fetch) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fetch, hipEventDisableTiming));
        // (_begin has synchronised the handle's stream: d_packed is complete; the copy runs on a stream of its own so
        // that the handle's next upload does not queue behind it)
        HIP_TRY(c, hipMemcpyAsync(out, c->d_packed, (size_t)c->packed_ready, hipMemcpyDeviceToHost, c->aux[0]));
        HIP_TRY(c, hipEventRecord(c->ev_fetch, c->aux[0]));
        c->fetch_pending = true;
    }
    c->packed_ready = 0;
    return FHIP_OK;
}

int fhip_frames_packed_fetch_wait(fhip_ctx *c)
{
    if (!c) return FHIP_E_INVALID;
    if (c->fetch_pending) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipEventSynchronize(c->ev_fetch));
        c->fetch_pending = false;
    }
    return FHIP_OK;
}

int fhip_encode_frames_packed(fhip_ctx *c, const fhip_batch *b, uint8_t *out, int64_t out_cap,
                              int64_t *out_bytes)
{
    if (!out || !out_bytes) return fail(c, FHIP_E_INVALID, "null argument");
    *out_bytes = 0;
    int64_t total = 0;
    int rc = fhip_frames_packed_begin(c, b, &total);
    if (rc != FHIP_OK) return rc;
    rc = fhip_frames_packed_fetch(c, out, out_cap);
    if (rc != FHIP_OK) return rc;
    *out_bytes = total;
    return FHIP_OK;
}

// ---- variable block size: encode_frame_vbs (vbs.c:85-119) around encode_frame, batched --------
//
// Everything between the blocks' PCM and the packed stream stays on the device and on the
// handle's stream; the host never learns how the blocks were split.  split_frame_v1 (k_vbs_split),
// then k_vbs_plan: a piece is k eighths of its block, so there are eight bins of equal piece
// length, each with a fixed range of frame slots (VbsBins: host arithmetic on nblocks alone) and a
// count on the device.  The path runs once per bin -- grids sized for the bin's capacity, counts
// read from the device, the pieces encoded where they lie in the caller's blocks (no gather) --
// the bins fanned out over the handle's internal streams and joined again on every way out;
// k_pack_frames_perm then packs the frames of all bins in stream order.
namespace {

int vbs_bins(const fhip_ctx *c, int nblocks, int block_size, fhip::VbsBins *vb, long long *frames_bytes,
             long long *bits_bytes)
{
    const fhip_params &p = c->p;
    const long long nch = p.channels;
    long long slot0 = 0, smp = 0, fr = 0, bt = 0;
    for (int k = 0; k < 8; k++) {
        const int n = (k + 1) * (block_size / 8);
        vb->n[k] = n;
        vb->cap[k] = (8 / (k + 1)) * nblocks;
        vb->slot0[k] = (int)slot0;
        vb->smp_off[k] = smp;
        vb->stride[k] = fhip_frame_stride(&p, n);
        vb->slot[k] = (vb->stride[k] + 3) & ~(long long)3;
        vb->fr_off[k] = fr;
        vb->bits_off[k] = bt;
        slot0 += vb->cap[k];
        smp += (long long)vb->cap[k] * n * nch;
        fr += (long long)vb->cap[k] * vb->stride[k];
        bt += (long long)vb->cap[k] * nch * vb->slot[k];
    }
    if ((size_t)slot0 > c->ws_frames || (size_t)smp > (size_t)c->max_frames * (size_t)nch * (size_t)p.block_size)
        return FHIP_E_INVALID;
    *frames_bytes = fr;
    *bits_bytes = bt;
    return FHIP_OK;
}

// joins the aux streams back into the caller's stream when the scope ends, however it ends
struct FanJoin {
    fhip_ctx *c; int nstreams; bool forked = false;
    FanJoin(fhip_ctx *cc, int ns) : c(cc), nstreams(ns) {}
    hipError_t fork()
    {
        hipError_t e = hipEventRecord(c->ev_fork, c->stream);
        for (int h = 0; h < nstreams && e == hipSuccess; h++) e = hipStreamWaitEvent(c->aux[h], c->ev_fork, 0);
        forked = true;                      // even partly: join whatever was reached
        return e;
    }
    hipError_t join()
    {
        if (!forked) return hipSuccess;
        forked = false;
        hipError_t first = hipSuccess;
        for (int h = 0; h < nstreams; h++) {
            hipError_t e = hipEventRecord(c->ev_join[h], c->aux[h]);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join[h], 0);
            if (e != hipSuccess) {              // last resort: nothing of the failed call stays in flight
                (void)hipStreamSynchronize(c->aux[h]);
                if (first == hipSuccess) first = e;
            }
        }
        return first;
    }
    ~FanJoin() { (void)join(); }
};

struct VbsOut {
    uint8_t *packed; long long cap;           // device
    int32_t *stream_bytes, *block_bytes, *block_frames;   // device, optional
    long long *totals;                        // device [4]
};

// pcm: DEVICE blocks.  Asynchronous on the handle's stream.
int vbs_dev_core(fhip_ctx *c, const int32_t *pcm, int nblocks, int block_size, uint32_t first_frame_number,
                 const VbsOut &o)
{
    const fhip_params &p = c->p;
    const size_t nch = (size_t)p.channels;
    fhip::VbsBins vb;
    long long frames_bytes = 0, bits_bytes = 0;
    if (vbs_bins(c, nblocks, block_size, &vb, &frames_bytes, &bits_bytes) != FHIP_OK)
        return fail(c, FHIP_E_INVALID, "nblocks * 8 exceeds the handle's max_frames");
    int rc = ensure_staging(c, 0);
    if (rc != FHIP_OK) return rc;
    const size_t slots = c->ws_frames, nstream = (size_t)c->max_frames;
    if (!c->d_srcoff) HIP_TRY(c, hipMalloc((void **)&c->d_srcoff, slots * sizeof(long long)));
    if (!c->d_frame_src) HIP_TRY(c, hipMalloc((void **)&c->d_frame_src, slots * sizeof(long long)));
    if (!c->d_fbytes) HIP_TRY(c, hipMalloc((void **)&c->d_fbytes, slots * sizeof(int32_t)));
    if (!c->d_fnum) HIP_TRY(c, hipMalloc((void **)&c->d_fnum, slots * sizeof(uint32_t)));
    if (!c->d_order) HIP_TRY(c, hipMalloc((void **)&c->d_order, nstream * sizeof(int32_t)));
    if (!c->d_offsets) HIP_TRY(c, hipMalloc((void **)&c->d_offsets, (nstream + 1) * sizeof(long long)));
    if (!c->d_vcnt) HIP_TRY(c, hipMalloc((void **)&c->d_vcnt, fhip::VBS_CNT_WORDS * sizeof(int32_t)));
    if (!c->d_first) HIP_TRY(c, hipMalloc((void **)&c->d_first, (nstream / 8 + 2) * sizeof(int32_t)));
    if ((size_t)bits_bytes > c->d_bits_bytes) {
        if (c->d_bits) (void)hipFree(c->d_bits);
        c->d_bits = nullptr; c->d_bits_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_bits, (size_t)bits_bytes));
        c->d_bits_bytes = (size_t)bits_bytes;
    }
    if ((size_t)frames_bytes > c->d_frames_bytes) {
        if (c->d_frames) (void)hipFree(c->d_frames);
        c->d_frames = nullptr; c->d_frames_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_frames, (size_t)frames_bytes));
        c->d_frames_bytes = (size_t)frames_bytes;
    }

    // ---- split_frame_v1 (vbs.c:36-83) and the piece tables, on the device ----
    // (d_opt / d_shift are free until a bin's K2 runs, which the plan precedes on this stream)
    HIP_TRY(c, fhip::launch_vbs_split(c->stream, pcm, nblocks, block_size, p.channels, c->d_opt, c->d_shift));
    HIP_TRY(c, fhip::launch_vbs_plan(c->stream, c->d_opt, c->d_shift, nblocks, block_size, p.channels,
                                     first_frame_number, vb, c->d_vcnt, c->d_order, c->d_frame_src, c->d_srcoff,
                                     c->d_fnum, c->d_first));

    // ---- one pass of the path per bin ----
    // A bin is a few hundred frames -- too few workgroups to fill the chip, and eight bins one
    // behind the other pay eight times the latency of the path's launches: the bins touch disjoint
    // ranges of every buffer and fan out over the handle's internal streams (measured per 1024
    // blocks, levels 9 / 10 / 12: one stream 2.7 / 2.8 / 5.4 ms, two 2.1 / 2.3 / 4.7, three 2.0 / 2.2 /
    // 4.2, four 2.2 / 2.1 / 4.3), longest pieces first, each to the stream with least capacity queued.
    static const bool vbs_serial = getenv("FHIP_VBS_SERIAL") != nullptr;          // measurements only
    constexpr int NA = fhip_ctx::NAUX;
    // lanes: the handle's own stream and vbs_lanes - 1 internal ones.  (Round 4: the kernel trace of a level-10 batch showed
    // three internal streams on TWO hardware queues -- sixteen part-filled launches one after another, 350 of 585 us -- and
    // the handle's own queue idle meanwhile.)
    static const int vbs_lanes = getenv("FHIP_VBS_STREAMS") ? std::max(2, std::min(NA + 1, atoi(getenv("FHIP_VBS_STREAMS")))) : 3;
    const int vbs_streams = vbs_lanes - 1;
    bool fan = !vbs_serial && !c->profiling;
    for (int h = 0; h < vbs_streams; h++) fan = fan && c->aux[h];
    FanJoin fj(c, vbs_streams);
    const int hint_frames = std::max(1, nblocks / 2);
    const int32_t *cnt_frames = c->d_vcnt + fhip::VBS_CNT_FRAMES, *cnt_sub = c->d_vcnt + fhip::VBS_CNT_SUB;
    struct MaybeProf {
        Prof *p;
        MaybeProf(fhip_ctx *cc, bool on, int i) : p(on ? new Prof(cc, i) : nullptr) {}
        ~MaybeProf() { delete p; }
    };
    // the lane a bin's launches go to: the one with the least estimated work queued.  A bin's search + K3 is a launch's
    // latency plus its live pieces' work, and the short pieces' bins hold more pieces (measured per 1024 blocks, eighths
    // 1 .. 8, with three lanes busy: level 10 -- the vector searches -- 175, 195, 192, 56, 54, 39, 72, 40 us; level 12 -- the
    // matrix searches -- 275, 265, 245, 168, 137, 136, 115, 113): the weights are those, in units of 5 us.
    static const int bin_weight_vec[8] = {35, 39, 38, 11, 11, 8, 14, 8}, bin_weight_mat[8] = {55, 53, 49, 34, 27, 27, 23, 23};
    const bool mat_search = p.order_method == 5 && p.bits_per_sample <= 24 && (p.bits_per_sample > 16 || p.max_prediction_order > 16);
    const int *bin_weight = mat_search ? bin_weight_mat : bin_weight_vec;
    long long queued[NA + 1] = {0};
    auto pick_stream = [&](int k) -> hipStream_t {
        if (!fan) return c->stream;
        int h = 0;
        for (int q = 1; q < vbs_lanes; q++) if (queued[q] < queued[h]) h = q;
        queued[h] += bin_weight[k];
        return h == 0 ? c->stream : c->aux[h - 1];
    };

    // K1, K2 and K4 do not depend on the piece length: ONE launch each over all bins (kernels.h:
    // MultiBin) where every bin is whole K1 tiles; K0, the order search and K3 run per bin
    const bool lpc_path = p.prediction_type == 2;
    if (lpc_path && fhip::autocorr_bins_supported(p.max_prediction_order, vb.n, 8)) {
        const bool prof = c->profiling;
        bool narrow[8];
        // ---- K0 per bin, back to back on the handle's stream (eight launches of ~10 us; fanned over the
        // internal streams they ran in 55 us but the fork and the join around them cost 60: the kernel
        // trace of a level-12 batch, round 3) ----
        static const bool k0_fan = getenv("FHIP_VBS_K0_FAN") != nullptr;                 // measurements only
        static const bool k0_per_bin = getenv("FHIP_VBS_K0_PER_BIN") != nullptr;         // measurements only
        for (int k = 0; k < 8; k++) narrow[k] = fhip::narrow_rows_ok(p, hint_frames * (int)nch, vb.n[k], true, true);
        if (!k0_per_bin && !k0_fan && fhip::prepare_bins_supported(p, vb.n, 8)) {
            // stereo: one launch over all bins, the longest pieces first
            fhip::MultiBin m0{};
            m0.nbins = 8;
            m0.cnt = cnt_frames;
            int wg = 0;
            for (int j = 0; j < 8; j++) {
                const int k = 7 - j;
                m0.cnt_ix[j] = k;
                m0.wg0[j] = wg;
                wg += fhip::prepare_bins_workgroups(vb.n[k], vb.cap[k], vb.n[7]);
                m0.n[j] = vb.n[k];
                m0.unit0[j] = vb.slot0[k];
                m0.cap[j] = vb.cap[k];
                m0.narrow[j] = narrow[k] ? 1 : 0;
                m0.smp_off[j] = vb.smp_off[k];
            }
            m0.wg0[8] = wg;
            MaybeProf pr(c, prof, 0);
            HIP_TRY(c, fhip::launch_prepare_bins(c->stream, p, pcm, m0, c->d_smp, c->d_k0rec, c->d_frame_src));
        } else {
            if (fan && k0_fan) HIP_TRY(c, fj.fork());
            for (int k = 7; k >= 0; k--) {
                const size_t sub0 = (size_t)vb.slot0[k] * nch;
                MaybeProf pr(c, prof, 0);
                HIP_TRY(c, fhip::launch_prepare((fan && k0_fan) ? pick_stream(k) : c->stream, p, pcm, vb.cap[k], vb.n[k],
                                                c->d_smp + vb.smp_off[k], c->d_k0rec + sub0, false, narrow[k],
                                                c->d_frame_src + vb.slot0[k], cnt_frames + k));
            }
            HIP_TRY(c, fj.join());
        }
        // ---- K1 (+ K2) over all bins, longest chains first ----
        const bool lpc_tail = p.max_prediction_order <= 12 && getenv("FHIP_NO_LPC_TAIL") == nullptr;
        fhip::MultiBin m1{};
        m1.nbins = 8;
        m1.cnt = cnt_sub;
        int wg = 0;
        for (int j = 0; j < 8; j++) {
            const int k = 7 - j;
            m1.cnt_ix[j] = k;
            m1.wg0[j] = wg;
            wg += (vb.cap[k] * (int)nch + 31) / 32;
            m1.n[j] = vb.n[k];
            m1.unit0[j] = vb.slot0[k] * (int)nch;
            m1.cap[j] = vb.cap[k] * (int)nch;
            m1.narrow[j] = narrow[k] ? 1 : 0;
            m1.smp_off[j] = vb.smp_off[k];
            m1.c[j] = (2.0 / (vb.n[k] - 1.0)) - 1.0;                  // lpc.c:34, on the host as launch_autocorr does
        }
        m1.wg0[8] = wg;
        const fhip::autocorr_lpc_out lo{p.lpc_precision, p.order_method, c->d_coefs, c->d_shift, c->d_opt, c->d_fin};
        {
            // (orders above 12: the bins of six eighths and more as a launch of their own with two workgroups
            // per tile -- even / odd lags, five chains per wave instead of nine -- beside the other bins' took
            // K1 from 184 to 131 us in the kernel trace and the batch from 1.245 to 1.23 ms: within the
            // noise, for a second fork / join; not kept)
            MaybeProf pr(c, prof, 1);
            HIP_TRY(c, fhip::launch_autocorr_bins(c->stream, m1, c->d_smp, p.max_prediction_order, c->d_autoc,
                                                  c->d_k0rec, lpc_tail ? &lo : nullptr));
        }
        fhip::MultiBin m2{};                                          // bins in slot order: K2, K4
        m2.nbins = 8;
        for (int k = 0; k < 8; k++) {
            m2.cnt_ix[k] = k;
            m2.wg0[k] = vb.slot0[k];
            m2.n[k] = vb.n[k];
            m2.cap[k] = vb.cap[k];
            m2.unit0[k] = vb.slot0[k];
            m2.stride[k] = vb.stride[k]; m2.fr_off[k] = vb.fr_off[k];
            m2.slot[k] = vb.slot[k]; m2.bits_off[k] = vb.bits_off[k];
            const int bps = p.bits_per_sample;
            m2.vsize[k] = (p.channels == 2) ? 16 + ((vb.n[k] * (bps + bps + 1) + 7) >> 3)
                                            : 16 + ((vb.n[k] * p.channels * bps + 7) >> 3);
        }
        m2.wg0[8] = vb.slot0[7] + vb.cap[7];
        if (!lpc_tail) {
            fhip::MultiBin mk = m2;                                   // K2's units are subframes
            mk.cnt = cnt_sub;
            for (int k = 0; k < 8; k++) { mk.unit0[k] = vb.slot0[k] * (int)nch; mk.cap[k] = vb.cap[k] * (int)nch; }
            MaybeProf pr(c, prof, 2);
            HIP_TRY(c, fhip::launch_lpc_bins(c->stream, mk, c->d_autoc, p.max_prediction_order, p.lpc_precision,
                                             p.order_method, c->d_coefs, c->d_shift, c->d_opt, c->d_fin));
        }
        // ---- order search + K3: a launch each per bin; the thinly filled bins of a small batch -- four eighths and longer:
        // a few hundred pieces each, whose launches are latency one after another -- grouped by the kernel they can share
        // (k_order_search_bins / k_encode_bins).  Units (a bin, or a group) go to the lanes by estimated work, heaviest first.
        if (fan) HIP_TRY(c, fj.fork());
        for (int q = 0; q <= NA; q++) queued[q] = 0;
        hipStream_t lane_of[8];
        static const int merge_max = getenv("FHIP_VBS_MERGE_MAX") ? atoi(getenv("FHIP_VBS_MERGE_MAX")) : 2048;   // blocks
        int unit_of[8], nunits = 0, unit_w[8] = {0};
        {
            int gkey[8];
            for (int k = 0; k < 8; k++) {
                const bool sr = fhip::order_search_supported(p, vb.n[k]);
                const int sg = sr ? fhip::order_search_group(p, vb.n[k]) : -1;
                // (the 256-thread geometries of runs up to 16 only: the two 128-thread bins of runs of 20 and 28 took 84 us in one
                // launch, 40 and 60 on different lanes)
                gkey[k] = (nblocks <= merge_max && k >= 3 && sg == 256 && fhip::encode_group(p, vb.n[k], true) == 256) ? sg : -1;
                unit_of[k] = -1;
                for (int q = 3; q < k && gkey[k] >= 0; q++) if (gkey[q] == gkey[k]) { unit_of[k] = unit_of[q]; break; }
                if (unit_of[k] < 0) unit_of[k] = nunits++;
                // (a group costs its first member's latency once: 95 us of three launches were 55 in one)
                unit_w[unit_of[k]] += (unit_w[unit_of[k]] > 0) ? 2 : bin_weight[k];
            }
        }
        bool unit_done[8] = {false};
        for (int turn = 0; turn < nunits; turn++) {
            int u = -1;
            for (int q = 0; q < nunits; q++) if (!unit_done[q] && (u < 0 || unit_w[q] > unit_w[u])) u = q;
            unit_done[u] = true;
            // the lane with the least estimated work queued
            hipStream_t st = c->stream;
            if (fan) {
                int h = 0;
                for (int q = 1; q < vbs_lanes; q++) if (queued[q] < queued[h]) h = q;
                queued[h] += unit_w[u];
                st = h == 0 ? c->stream : c->aux[h - 1];
            }
            int members = 0, first = -1;
            for (int k = 0; k < 8; k++) if (unit_of[k] == u) { lane_of[k] = st; members++; if (first < 0) first = k; }
            if (members > 1) {
                fhip::MultiBin mg{};
                mg.cnt = cnt_sub;
                int nb = 0, wgs = 0;
                for (int k = 0; k < 8; k++) {
                    if (unit_of[k] != u) continue;
                    mg.cnt_ix[nb] = k;
                    mg.wg0[nb] = wgs;
                    wgs += vb.cap[k] * (int)nch;
                    mg.n[nb] = vb.n[k];
                    mg.unit0[nb] = vb.slot0[k] * (int)nch;
                    mg.cap[nb] = vb.cap[k] * (int)nch;
                    mg.narrow[nb] = narrow[k] ? 1 : 0;
                    mg.smp_off[nb] = vb.smp_off[k];
                    mg.slot[nb] = vb.slot[k];
                    mg.bits_off[nb] = vb.bits_off[k];
                    nb++;
                }
                mg.nbins = nb;
                mg.wg0[nb] = wgs;
                {
                    MaybeProf pr(c, prof, 5);
                    HIP_TRY(c, fhip::launch_order_search_bins(st, p, mg, c->d_smp, c->d_coefs, c->d_shift, c->d_opt, c->d_fin,
                                                              c->d_k0rec));
                }
                MaybeProf pr(c, prof, 3);
                HIP_TRY(c, fhip::launch_encode_bins(st, p, mg, c->d_smp, c->d_coefs, c->d_shift, c->d_opt, c->d_fin, c->d_info,
                                                    c->d_k0rec, c->d_bits));
                continue;
            }
            const int k = first;
            const int n = vb.n[k];
            const size_t sub0 = (size_t)vb.slot0[k] * nch;
            const int nsub_cap = vb.cap[k] * (int)nch;
            int32_t *coefs = c->d_coefs + sub0 * FHIP_MAX_ORDER * FHIP_MAX_ORDER;
            int32_t *shift = c->d_shift + sub0 * FHIP_MAX_ORDER;
            int32_t *opt = c->d_opt + sub0, *fin = c->d_fin + sub0 * fhip::FIN_STRIDE;
            const bool searched = fhip::order_search_supported(p, n);
            if (searched) {
                MaybeProf pr(c, prof, 5);
                HIP_TRY(c, fhip::launch_order_search(st, p, c->d_smp + vb.smp_off[k], nsub_cap, n, coefs, shift, opt,
                                                     fin, c->d_k0rec + sub0, narrow[k], cnt_sub + k));
            }
            MaybeProf pr(c, prof, 3);
            HIP_TRY(c, fhip::launch_encode(st, p, c->d_smp + vb.smp_off[k], nsub_cap, n, coefs, shift, opt, fin,
                                           c->d_info + sub0, nullptr, c->d_bits + vb.bits_off[k], vb.slot[k], -1, 0,
                                           narrow[k], c->d_k0rec + sub0, searched, cnt_sub + k));
        }
        // ---- K4: one launch per lane over the lane's bins, behind their K3s (round 4: one launch over all bins behind
        // the join left it, and the 18 us a cross-queue wait takes, on the critical path while lanes idled) ----
        m2.cnt = cnt_frames;
        {
            hipStream_t done[8];
            int ndone = 0;
            for (int k = 0; k < 8; k++) {
                bool seen = false;
                for (int q = 0; q < ndone; q++) seen = seen || done[q] == lane_of[k];
                if (seen) continue;
                done[ndone++] = lane_of[k];
                fhip::MultiBin ml = m2;
                int nb = 0, wgs = 0;
                for (int q = 0; q < 8; q++) {
                    if (lane_of[q] != lane_of[k]) continue;
                    ml.cnt_ix[nb] = q;
                    ml.wg0[nb] = wgs;
                    wgs += vb.cap[q];
                    ml.n[nb] = m2.n[q]; ml.cap[nb] = m2.cap[q]; ml.unit0[nb] = m2.unit0[q];
                    ml.stride[nb] = m2.stride[q]; ml.fr_off[nb] = m2.fr_off[q];
                    ml.slot[nb] = m2.slot[q]; ml.bits_off[nb] = m2.bits_off[q]; ml.vsize[nb] = m2.vsize[q];
                    nb++;
                }
                ml.nbins = nb;
                ml.wg0[nb] = wgs;
                MaybeProf pr(c, prof, 4);
                HIP_TRY(c, fhip::launch_assemble_bins(lane_of[k], p, ml, pcm, c->d_info, c->d_bits, c->d_frames, c->d_fbytes,
                                                      c->d_fnum, c->d_frame_src));
            }
        }
        HIP_TRY(c, fj.join());
    } else {
        if (fan) HIP_TRY(c, fj.fork());
        for (int k = 7; k >= 0; k--) {
            const int n = vb.n[k];
            const size_t sub0 = (size_t)vb.slot0[k] * nch;
            const FrameOut fo{c->d_frames + vb.fr_off[k], vb.stride[k], c->d_fbytes + vb.slot0[k], 0,
                              c->d_fnum + vb.slot0[k]};
            const Ragged rg{c->d_frame_src + vb.slot0[k], cnt_frames + k, cnt_sub + k, hint_frames};
            rc = run_range(c, pick_stream(k), c->profiling && !fan, pcm, vb.cap[k], n, c->d_info + sub0, nullptr,
                           c->d_bits + vb.bits_off[k], vb.slot[k], c->d_smp + vb.smp_off[k],
                           c->d_autoc + sub0 * FHIP_MAX_LAGS, sub0, fo, false, nullptr, false, &rg);
            if (rc != FHIP_OK) return rc;                       // (~FanJoin joins what was queued)
        }
        HIP_TRY(c, fj.join());
    }
    HIP_TRY(c, fhip::launch_pack_frames_perm(c->stream, c->d_frames, c->d_srcoff, c->d_fbytes, c->d_order,
                                             8 * nblocks, c->d_vcnt + fhip::VBS_CNT_ALL, c->d_offsets,
                                             o.packed, o.cap, o.stream_bytes, o.totals));
    HIP_TRY(c, fhip::launch_vbs_block_bytes(c->stream, c->d_first, c->d_offsets, nblocks, o.block_bytes,
                                            o.block_frames));
    return FHIP_OK;
}

int vbs_check(fhip_ctx *c, const void *pcm, int nblocks, int block_size)
{
    if (!c || !pcm) return fail(c, FHIP_E_INVALID, "null argument");
    const fhip_params &p = c->p;
    if (!p.variable_block_size || !p.allow_vbs)
        return fail(c, FHIP_E_INVALID, "the handle's parameters have no variable block size");
    if (block_size > p.block_size || block_size < 128 || (block_size % 8))
        return fail(c, FHIP_E_INVALID, "vbs needs block_size % 8 == 0 and >= 128 (vbs.c:93)");
    if (nblocks < 0 || (long long)nblocks * 8 > c->max_frames)
        return fail(c, FHIP_E_INVALID, "nblocks * 8 exceeds the handle's max_frames");
    return FHIP_OK;
}

// the device entry reads the caller's blocks where they lie, with 16-byte loads (k_vbs_split, K0)
int vbs_check_dev_pcm(fhip_ctx *c, const void *pcm)
{
    if (reinterpret_cast<uintptr_t>(pcm) & 15)
        return fail(c, FHIP_E_INVALID, "device pcm must be 16-byte aligned (flakehip.h: fhip_encode_blocks_vbs_dev)");
    return FHIP_OK;
}

}  // namespace

int fhip_encode_blocks_vbs_dev(fhip_ctx *c, const int32_t *pcm, int nblocks, int block_size,
                               uint32_t first_frame_number, const fhip_vbs_out *out)
{
    int rc = vbs_check(c, pcm, nblocks, block_size);
    if (rc == FHIP_OK) rc = vbs_check_dev_pcm(c, pcm);
    if (rc != FHIP_OK) return rc;
    if (!out || !out->packed || !out->totals || out->packed_cap < 0)
        return fail(c, FHIP_E_INVALID, "null output argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (nblocks == 0) {
        HIP_TRY(c, hipMemsetAsync(out->totals, 0, 4 * sizeof(int64_t), c->stream));
        return FHIP_OK;
    }
    return vbs_dev_core(c, pcm, nblocks, block_size, first_frame_number,
                        VbsOut{out->packed, (long long)out->packed_cap, out->frame_bytes, out->block_bytes,
                               out->block_frames, reinterpret_cast<long long *>(out->totals)});
}

int fhip_encode_blocks_vbs_packed(fhip_ctx *c, const int32_t *pcm, int nblocks, int block_size,
                                  uint32_t first_frame_number, uint8_t *out, int64_t out_cap,
                                  int32_t *block_bytes, int32_t *block_frames, int64_t *out_bytes,
                                  int32_t *max_frame_bytes, uint32_t *next_frame_number)
{
    int rc = vbs_check(c, pcm, nblocks, block_size);
    if (rc != FHIP_OK) return rc;
    if (!out || !out_bytes || !block_bytes) return fail(c, FHIP_E_INVALID, "null argument");
    const fhip_params &p = c->p;
    *out_bytes = 0;
    c->packed_ready = 0;                     // d_packed is about to be rewritten (and may move)
    if (max_frame_bytes) *max_frame_bytes = 0;
    if (next_frame_number) *next_frame_number = first_frame_number;
    if (nblocks == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = ensure_staging(c, 0);
    if (rc != FHIP_OK) return rc;
    fhip::VbsBins vb;
    long long frames_bytes = 0, bits_bytes = 0;
    if (vbs_bins(c, nblocks, block_size, &vb, &frames_bytes, &bits_bytes) != FHIP_OK)
        return fail(c, FHIP_E_INVALID, "nblocks * 8 exceeds the handle's max_frames");
    if ((size_t)frames_bytes > c->d_packed_bytes) {
        if (c->d_packed) (void)hipFree(c->d_packed);
        c->d_packed = nullptr; c->d_packed_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_packed, (size_t)frames_bytes));
        c->d_packed_bytes = (size_t)frames_bytes;
    }
    const size_t nb = (size_t)c->max_frames / 8 + 2;
    if (!c->d_totals) HIP_TRY(c, hipMalloc((void **)&c->d_totals, 4 * sizeof(long long)));
    if (!c->d_blk_bytes) HIP_TRY(c, hipMalloc((void **)&c->d_blk_bytes, nb * sizeof(int32_t)));
    if (!c->d_blk_frames) HIP_TRY(c, hipMalloc((void **)&c->d_blk_frames, nb * sizeof(int32_t)));

    // one upload, the batch on the device, one download of the stream's bytes
    const size_t nvals = (size_t)nblocks * block_size * (size_t)p.channels;
    HIP_TRY(c, hipMemcpyAsync(c->d_pcm, pcm, nvals * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = vbs_dev_core(c, c->d_pcm, nblocks, block_size, first_frame_number,
                      VbsOut{c->d_packed, (long long)c->d_packed_bytes, nullptr, c->d_blk_bytes,
                             block_frames ? c->d_blk_frames : nullptr, c->d_totals});
    if (rc != FHIP_OK) return rc;
    long long totals[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(totals, c->d_totals, sizeof totals, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(block_bytes, c->d_blk_bytes, (size_t)nblocks * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (block_frames)
        HIP_TRY(c, hipMemcpyAsync(block_frames, c->d_blk_frames, (size_t)nblocks * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    rc = fhip_sync(c);
    if (rc != FHIP_OK) return rc;
    // every FRAME is checked (k_frame_offsets_perm, totals[3] bit 1), before anything is copied out: a piece that
    // failed inside a block of several pieces does not show in the block's byte count
    if (totals[3] & 2) return fail(c, FHIP_E_GENERIC, "a frame of the batch was not encoded");
    for (int b = 0; b < nblocks; b++)
        if (block_bytes[b] <= 0) return fail(c, FHIP_E_GENERIC, "a frame of the batch was not encoded");
    if (totals[1] > out_cap) return fail(c, FHIP_E_INVALID, "output buffer too small for the batch's frames");
    HIP_TRY(c, hipMemcpy(out, c->d_packed, (size_t)totals[1], hipMemcpyDeviceToHost));
    if (max_frame_bytes) *max_frame_bytes = (int32_t)totals[2];
    if (next_frame_number) *next_frame_number = first_frame_number + (uint32_t)((long long)nblocks * block_size);
    *out_bytes = totals[1];
    return FHIP_OK;
}

int fhip_prepare_frames(fhip_ctx *c, const int32_t *pcm, int nframes, int n,
                        int32_t *samples, fhip_subframe_info *info)
{
    if (!c || !pcm || !samples || !info) return fail(c, FHIP_E_INVALID, "null argument");
    if (nframes < 0 || nframes > c->max_frames || n < 1 || n > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "batch shape out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, 0);
    if (rc != FHIP_OK) return rc;
    const size_t nsub = (size_t)nframes * c->p.channels;
    if (nsub == 0) return FHIP_OK;
    HIP_TRY(c, hipMemcpyAsync(c->d_pcm, pcm, nsub * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, nsub * sizeof(fhip_subframe_info), c->stream));
    HIP_TRY(c, fhip::launch_prepare(c->stream, c->p, c->d_pcm, nframes, n, c->d_smp, c->d_info));
    HIP_TRY(c, hipMemcpyAsync(samples, c->d_smp, nsub * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(info, c->d_info, nsub * sizeof(fhip_subframe_info), hipMemcpyDeviceToHost, c->stream));
    return fhip_sync(c);
}

int fhip_lpc_calc_coefs(fhip_ctx *c, const int32_t *samples, int nsub, int n,
                        int max_order, int precision, int omethod,
                        int32_t *coefs, int32_t *shift, int32_t *opt_order, double *autoc)
{
    if (!c || !samples || !coefs || !shift || !opt_order) return fail(c, FHIP_E_INVALID, "null argument");
    const size_t cap = (size_t)c->max_frames * c->p.channels;
    if (nsub < 0 || (size_t)nsub > cap || n < 1 || n > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "batch shape out of range");
    if (max_order < 1 || max_order > FHIP_MAX_ORDER || n <= max_order)
        return fail(c, FHIP_E_INVALID, "max_order out of range (optimize.c:168)");
    if (omethod < 0 || omethod > 6 || precision < 2 || precision > 15)
        return fail(c, FHIP_E_INVALID, "bad order method / precision");
    if (nsub == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t ns = (size_t)nsub;
    HIP_TRY(c, hipMemcpyAsync(c->d_smp, samples, ns * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_coefs, 0, ns * FHIP_MAX_ORDER * FHIP_MAX_ORDER * sizeof(int32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_shift, 0, ns * FHIP_MAX_ORDER * sizeof(int32_t), c->stream));
    HIP_TRY(c, fhip::launch_autocorr(c->stream, c->d_smp, nsub, n, max_order, c->d_autoc));
    HIP_TRY(c, fhip::launch_lpc(c->stream, c->d_autoc, nsub, max_order, precision, omethod,
                                c->d_coefs, c->d_shift, c->d_opt, c->d_fin));
    HIP_TRY(c, hipMemcpyAsync(coefs, c->d_coefs, ns * FHIP_MAX_ORDER * FHIP_MAX_ORDER * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(shift, c->d_shift, ns * FHIP_MAX_ORDER * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(opt_order, c->d_opt, ns * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (autoc)
        HIP_TRY(c, hipMemcpyAsync(autoc, c->d_autoc, ns * FHIP_MAX_LAGS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return fhip_sync(c);
}

int fhip_encode_residual(fhip_ctx *c, const int32_t *samples, int nsub, int n,
                         fhip_subframe_info *info, int32_t *residual,
                         uint8_t *rice_bits, int64_t rice_slot_bytes)
{
    if (!c || !samples || !info) return fail(c, FHIP_E_INVALID, "null argument");
    const size_t cap = (size_t)c->max_frames * c->p.channels;
    if (nsub < 0 || (size_t)nsub > cap || n < 1 || n > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "batch shape out of range");
    if (n > FHIP_MAX_BLOCK) return fail(c, FHIP_E_UNSUPPORTED, "block_size above FHIP_MAX_BLOCK");
    if (rice_bits && (rice_slot_bytes < 4 || (rice_slot_bytes & 3)))
        return fail(c, FHIP_E_INVALID, "rice_slot_bytes must be a positive multiple of 4");
    if (nsub == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t ns = (size_t)nsub;
    const size_t bits_bytes = rice_bits ? ns * (size_t)rice_slot_bytes : 0;
    int rc = ensure_staging(c, bits_bytes);
    if (rc != FHIP_OK) return rc;
    const fhip_params &p = c->p;
    HIP_TRY(c, hipMemcpyAsync(c->d_smp, samples, ns * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_info, info, ns * sizeof(fhip_subframe_info), hipMemcpyHostToDevice, c->stream));
    if (p.prediction_type == 2 && n > p.max_prediction_order && n >= 5) {
        HIP_TRY(c, fhip::launch_autocorr(c->stream, c->d_smp, nsub, n, p.max_prediction_order, c->d_autoc));
        HIP_TRY(c, fhip::launch_lpc(c->stream, c->d_autoc, nsub, p.max_prediction_order,
                                    p.lpc_precision, p.order_method, c->d_coefs, c->d_shift, c->d_opt, c->d_fin));
    }
    if (rice_bits) HIP_TRY(c, hipMemsetAsync(c->d_bits, 0, bits_bytes, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_k0rec, c->d_info, ns * sizeof(fhip_subframe_info), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, fhip::launch_encode(c->stream, p, c->d_smp, nsub, n, c->d_coefs, c->d_shift, c->d_opt, c->d_fin,
                                   c->d_info, residual ? c->d_res : nullptr,
                                   rice_bits ? c->d_bits : nullptr, rice_slot_bytes, -1, 0, false, c->d_k0rec));
    HIP_TRY(c, hipMemcpyAsync(info, c->d_info, ns * sizeof(fhip_subframe_info), hipMemcpyDeviceToHost, c->stream));
    if (residual)
        HIP_TRY(c, hipMemcpyAsync(residual, c->d_res, ns * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (rice_bits)
        HIP_TRY(c, hipMemcpyAsync(rice_bits, c->d_bits, bits_bytes, hipMemcpyDeviceToHost, c->stream));
    return fhip_sync(c);
}

int fhip_order_search_bits(fhip_ctx *c, const int32_t *samples, int nsub, int n,
                           const fhip_subframe_info *info, uint32_t *bits)
{
    if (!c || !samples || !info || !bits) return fail(c, FHIP_E_INVALID, "null argument");
    const size_t cap = (size_t)c->max_frames * c->p.channels;
    if (nsub < 0 || (size_t)nsub > cap || n < 1 || n > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "batch shape out of range");
    const fhip_params &p = c->p;
    if (!fhip::order_search_supported(p, n))
        return fail(c, FHIP_E_UNSUPPORTED, "no order-search kernel for this method / block size (the search then runs inside K3)");
    if (nsub == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, 0);
    if (rc != FHIP_OK) return rc;
    const size_t ns = (size_t)nsub;
    HIP_TRY(c, hipMemcpyAsync(c->d_smp, samples, ns * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_k0rec, info, ns * sizeof(fhip_subframe_info), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, fhip::launch_autocorr(c->stream, c->d_smp, nsub, n, p.max_prediction_order, c->d_autoc));
    HIP_TRY(c, fhip::launch_lpc(c->stream, c->d_autoc, nsub, p.max_prediction_order,
                                p.lpc_precision, p.order_method, c->d_coefs, c->d_shift, c->d_opt, c->d_fin));
    // the residual workspace ([nsub][n] int32, n >= 512 here) holds the table
    uint32_t *table = reinterpret_cast<uint32_t *>(c->d_res);
    HIP_TRY(c, fhip::launch_order_search(c->stream, p, c->d_smp, nsub, n, c->d_coefs, c->d_shift, c->d_opt, c->d_fin,
                                         c->d_k0rec, false, nullptr, table));
    HIP_TRY(c, hipMemcpyAsync(bits, table, ns * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    return fhip_sync(c);
}

int fhip_vbs_split(fhip_ctx *c, const int32_t *pcm, int nblocks, int block_size,
                   int32_t *frames, int32_t *sizes)
{
    if (!c || !pcm || !frames || !sizes) return fail(c, FHIP_E_INVALID, "null argument");
    if (nblocks < 0 || nblocks > c->max_frames || block_size > c->p.block_size ||
        block_size < 128 || (block_size % 8))
        return fail(c, FHIP_E_INVALID, "vbs needs block_size % 8 == 0 and >= 128 (vbs.c:93)");
    if (nblocks == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, 0);
    if (rc != FHIP_OK) return rc;
    const size_t nvals = (size_t)nblocks * block_size * c->p.channels;
    // d_opt (>= max_frames ints) and d_shift (>= 32*max_frames ints) are free before K2 runs
    HIP_TRY(c, hipMemcpyAsync(c->d_pcm, pcm, nvals * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, fhip::launch_vbs_split(c->stream, c->d_pcm, nblocks, block_size, c->p.channels,
                                      c->d_opt, c->d_shift));
    HIP_TRY(c, hipMemcpyAsync(frames, c->d_opt, (size_t)nblocks * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sizes, c->d_shift, (size_t)nblocks * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    return fhip_sync(c);
}

int fhip_calc_rice_params(fhip_ctx *c, const int32_t *residual, int nsub, int n,
                          int pred_order, int lpc, int bps, int pmin, int pmax,
                          fhip_subframe_info *info, uint8_t *rice_bits, int64_t rice_slot_bytes)
{
    if (!c || !residual || !info) return fail(c, FHIP_E_INVALID, "null argument");
    const size_t cap = (size_t)c->max_frames * c->p.channels;
    if (nsub < 0 || (size_t)nsub > cap || n < 1 || n > c->p.block_size)
        return fail(c, FHIP_E_INVALID, "batch shape out of range");
    if (n > FHIP_MAX_BLOCK) return fail(c, FHIP_E_UNSUPPORTED, "block_size above FHIP_MAX_BLOCK");
    if (pred_order < 0 || pred_order > FHIP_MAX_ORDER || pred_order > n || pmin < 0 || pmax > 8 || pmin > pmax)
        return fail(c, FHIP_E_INVALID, "bad prediction / partition order (rice.c:116-118)");
    if (rice_bits && (rice_slot_bytes < 4 || (rice_slot_bytes & 3)))
        return fail(c, FHIP_E_INVALID, "rice_slot_bytes must be a positive multiple of 4");
    if (nsub == 0) return FHIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t ns = (size_t)nsub;
    const size_t bits_bytes = rice_bits ? ns * (size_t)rice_slot_bytes : 0;
    int rc = ensure_staging(c, bits_bytes);
    if (rc != FHIP_OK) return rc;
    fhip_params p = c->p;
    p.min_partition_order = pmin;
    p.max_partition_order = pmax;
    std::vector<fhip_subframe_info> seed(ns);
    std::memset(seed.data(), 0, ns * sizeof(fhip_subframe_info));
    for (auto &s : seed) s.obits = bps;
    HIP_TRY(c, hipMemcpyAsync(c->d_smp, residual, ns * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_info, seed.data(), ns * sizeof(fhip_subframe_info), hipMemcpyHostToDevice, c->stream));
    if (rice_bits) HIP_TRY(c, hipMemsetAsync(c->d_bits, 0, bits_bytes, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_k0rec, seed.data(), ns * sizeof(fhip_subframe_info), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, fhip::launch_encode(c->stream, p, c->d_smp, nsub, n, c->d_coefs, c->d_shift, c->d_opt, c->d_fin,
                                   c->d_info, nullptr, rice_bits ? c->d_bits : nullptr,
                                   rice_slot_bytes, pred_order, lpc ? 1 : 0, false, c->d_k0rec));
    HIP_TRY(c, hipMemcpyAsync(info, c->d_info, ns * sizeof(fhip_subframe_info), hipMemcpyDeviceToHost, c->stream));
    if (rice_bits)
        HIP_TRY(c, hipMemcpyAsync(rice_bits, c->d_bits, bits_bytes, hipMemcpyDeviceToHost, c->stream));
    rc = fhip_sync(c);
    return rc;
}

}  // extern "C"
