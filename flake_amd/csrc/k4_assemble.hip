// k4_assemble.hip -- K4: whole FLAC frames on the device (encode.c:718-764, :800-917,
// :949-964) and the VBS splitter (vbs.c:36-83).
#include "device_util.h"
#ifdef FHIP_STAMPS
FHIP_DEFINE_STAMP_READER(fhip_debug_read_stamps_k4)      // tools/stamps_k4.py: phases of k_assemble's workgroup 0
#endif

namespace fhip {
namespace {

// ---------------------------------------------------------------------------
// K4  k_assemble -- whole FLAC frames on the device
// ---------------------------------------------------------------------------
// One workgroup per frame turns the side information of K0..K3 and the packed
// residual sections into the finished frame: frame header + CRC-8
// (encode.c:718-764), subframe headers, warm-up samples, LPC header
// (encode.c:800-905), the residual sections appended bit for bit, byte
// alignment, CRC-16 (encode.c:907-917), and the verbatim fallback of
// encode.c:949-964 when the frame would exceed its verbatim size or a section
// did not fit its slot.
//
// The frame is a concatenation of a few bit strings ("segments"): the frame
// header and one small prefix per subframe are built serially in LDS by one
// thread each; the bodies are the residual slots in HBM (or, for VERBATIM
// subframes, the samples themselves).  Every output dword is then produced by
// exactly one thread from the segments that overlap it, so stores are plain and
// coalesced.  CRC-16 is linear: each thread takes the CRC of a contiguous byte
// chunk and the partial CRCs are merged in a log-step tree with the constants
// x^(8*chunk*2^j) mod P.
constexpr int ASM_MAX_SEG = 2 * FHIP_MAX_CH + 2;
constexpr int ASM_PREFIX_BYTES = 224;     // 8+33 header bits, 32 warm-ups of <= 32 bits, 9 + 32*15 coef bits

struct AsmSeg { int kind; int ch; int nbits; int dst; };   // kind: 0 LDS dwords, 1 rice slot, 2 verbatim samples (a frame is < 2^31 bits)

constexpr int ASM_PREFIX_WORDS = ASM_PREFIX_BYTES / 4;

// A bit string in LDS is kept as stream-order dwords (dword j = stream bits 32 j .. 32 j + 31, the first bit on top): a
// field of nb bits (1 .. 33; v's bits past 32 are zero: fields wider than 32 bits are zero-extended) is ORed in at any
// bit position by whichever lane holds it.  (Rounds 2-3 wrote a subframe's prefix by ONE thread, field after field, each
// warm-up sample and coefficient a global load the next waited for: ~10 us of a frame's ~35.)
__device__ __forceinline__ void or_field(uint32_t *w, int pos, int nb, uint32_t v)
{
    if (nb <= 0) return;
    const int j = pos >> 5, off = pos & 31;
    const unsigned long long x = (unsigned long long)v << (64 - off - nb);        // off + nb <= 64
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    if (hi) atomicOr(&w[j], hi);
    if (lo) atomicOr(&w[j + 1], lo);
}

__device__ __forceinline__ uint16_t crc16_mulmod(uint16_t a, uint16_t b)
{
    // a * b mod x^16 + x^15 + x^2 + 1 over GF(2)
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        if (r & 0x10000u) r ^= 0x18005u;
        if ((b >> i) & 1u) r ^= a;
    }
    return (uint16_t)r;
}

// x^(8 * 2^i) mod P, i = 0 .. 23 (compile-time: repeated squaring of x^8)
constexpr uint16_t crc16_mulmod_c(uint16_t a, uint16_t b)
{
    uint32_t r = 0;
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        if (r & 0x10000u) r ^= 0x18005u;
        if ((b >> i) & 1u) r ^= a;
    }
    return (uint16_t)r;
}
struct Crc16Pow { uint16_t v[24]; };
constexpr Crc16Pow crc16_pow_table()
{
    Crc16Pow t{};
    uint16_t p = 0x100;
    for (int i = 0; i < 24; i++) { t.v[i] = p; p = crc16_mulmod_c(p, p); }
    return t;
}
__device__ __forceinline__ uint16_t crc16_pow8(int i)
{
    constexpr Crc16Pow t = crc16_pow_table();
    // (a wave-uniform index into 24 constants: a select chain, no memory)
    uint16_t r = t.v[0];
#pragma unroll
    for (int q = 1; q < 24; q++) r = (i == q) ? t.v[q] : r;
    return r;
}

// a * B mod P for a constant B: linear in a, the XOR of (x^i B mod P) over a's set bits -- three instructions a bit
struct Crc16Lin { uint16_t c[16]; };
constexpr Crc16Lin crc16_lin_table(uint16_t b)
{
    Crc16Lin t{};
    for (int i = 0; i < 16; i++) t.c[i] = crc16_mulmod_c(b, (uint16_t)(1u << i));
    return t;
}
template <uint16_t B>
__device__ __forceinline__ uint32_t crc16_mul_const(uint32_t a)
{
    constexpr Crc16Lin t = crc16_lin_table(B);
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= (uint32_t)((int32_t)(a << (31 - i)) >> 31) & (uint32_t)t.c[i];
    return r;
}
// crc * x^(8 e) mod P for e's bits JLO .. J: the CRC of a string followed by e more bytes, less those bytes' own CRC
template <int J, int JLO>
__device__ __forceinline__ uint32_t crc16_shift_bytes(uint32_t crc, int e)
{
    if constexpr (J >= JLO) {
        constexpr Crc16Pow t = crc16_pow_table();
        const uint32_t m = crc16_mul_const<t.v[J]>(crc);
        crc = ((e >> J) & 1) ? m : crc;
        return crc16_shift_bytes<J - 1, JLO>(crc, e);
    } else {
        return crc;
    }
}
// x^e mod P; x^32767 = 1 (P = (x + 1)(x^15 + x + 1), the second factor primitive), so x^(32767 - 8 k) undoes k bytes
constexpr uint16_t crc16_xpow_c(uint32_t e)
{
    uint16_t r = 1, b = 2;
    while (e) {
        if (e & 1u) r = crc16_mulmod_c(r, b);
        b = crc16_mulmod_c(b, b);
        e >>= 1;
    }
    return r;
}
static_assert(crc16_mulmod_c(crc16_xpow_c(8), crc16_xpow_c(32767 - 8)) == 1, "x^-8 mod the CRC-16 polynomial");
// the CRC of a string from the CRC of the string followed by e < 16 zero bytes
__device__ __forceinline__ uint32_t crc16_unshift_bytes(uint32_t crc, int e)
{
    const uint32_t m0 = crc16_mul_const<crc16_xpow_c(32767 - 8)>(crc);
    crc = (e & 1) ? m0 : crc;
    const uint32_t m1 = crc16_mul_const<crc16_xpow_c(32767 - 16)>(crc);
    crc = (e & 2) ? m1 : crc;
    const uint32_t m2 = crc16_mul_const<crc16_xpow_c(32767 - 32)>(crc);
    crc = (e & 4) ? m2 : crc;
    const uint32_t m3 = crc16_mul_const<crc16_xpow_c(32767 - 64)>(crc);
    return (e & 8) ? m3 : crc;
}

__device__ __forceinline__ int32_t asm_sample(const int32_t *pcm_frame, int nch, int ch, int t,
                                             int ch_mode, int wasted)
{
    // FlacSubframe.samples recomputed (encode.c:648-694, :558-593)
    int32_t v;
    if (nch == 2 && ch_mode != FHIP_CH_LEFT_RIGHT) {
        const int32_t l = pcm_frame[2 * t], r = pcm_frame[2 * t + 1];
        const int32_t side = (int32_t)((uint32_t)l - (uint32_t)r);
        if (ch_mode == FHIP_CH_MID_SIDE) v = ch ? side : ((int32_t)((uint32_t)l + (uint32_t)r) >> 1);
        else if (ch_mode == FHIP_CH_LEFT_SIDE) v = ch ? side : l;
        else v = ch ? r : side;
    } else {
        v = pcm_frame[(size_t)t * nch + ch];
    }
    return v >> wasted;
}

// Round 4: ONE WAVE per frame.  The stamps of a four-wave workgroup (tools/stamps_k4.py) showed a frame's 27 k cycles
// to be waits -- the record's loads, a thread's serial sums, the header's table chain, a memory round trip for every 256
// quads, the barriers between -- with eight frames in flight per CU; a wave per frame keeps 32 in flight, needs no
// barrier (the LDS is the wave's own), takes the per-channel sums by lanes and keeps four quads' loads in the air.
// (Also tried: a grid of what the chip holds at once walking the frames in stream order -- no workgroup for the slots of
// a ragged batch that stay empty, tables built once: 367 us for the frames of 8192 blocks where a workgroup per slot
// took 324; the same form lost 11-15 % in the order search.  The hardware's dispatch of a fresh workgroup balances
// better than a fixed walk, and a workgroup that finds its slot empty costs little.)
constexpr int AT = WAVE;                  // threads per frame
constexpr int ASM_QB = 4;                 // quads a lane keeps in flight
__global__ __launch_bounds__(AT)
void k_assemble(fhip_params P, int n, const int32_t *__restrict__ pcm,
                const fhip_subframe_info *__restrict__ info, const uint8_t *__restrict__ rice,
                long long slot_bytes, uint8_t *__restrict__ frames, long long frame_stride,
                int32_t *__restrict__ frame_bytes, uint32_t number_base, uint32_t number_step,
                const uint32_t *__restrict__ numbers,
                int sr_code0, int sr_code1, int bps_code, int verbatim_size,
                const long long *__restrict__ frame_src, const int32_t *__restrict__ dev_frames, MultiBin mb)
{
    if (dev_frames && (int)blockIdx.x >= dev_count(dev_frames, 0)) return;      // (a ragged batch's grid is its bin's capacity)
    int f = blockIdx.x;
    if (mb.nbins) {
        // several bins of a ragged batch in one launch (kernels.h: MultiBin; one workgroup per frame slot of the
        // bins listed): info / frame_bytes / numbers / frame_src are the handle's whole slot-indexed arrays, the
        // sections and the frames lie bin by bin
        const int k = find_bin(mb, f);
        const int local = f - mb.wg0[k];
        if (local >= __builtin_amdgcn_readfirstlane(mb.cnt[mb.cnt_ix[k]])) return;
        f = mb.unit0[k] + local;
        n = mb.n[k];
        verbatim_size = mb.vsize[k];
        slot_bytes = mb.slot[k];
        frame_stride = mb.stride[k];
        rice += mb.bits_off[k] - (long long)mb.unit0[k] * P.channels * slot_bytes;     // indexed by the global slot below
        frames += mb.fr_off[k] - (long long)mb.unit0[k] * frame_stride;
    }
    __shared__ uint32_t s_bits[FHIP_MAX_CH + 1][ASM_PREFIX_WORDS];   // row 0: the frame header; row c + 1: subframe c's prefix
    __shared__ AsmSeg s_seg[ASM_MAX_SEG];        // the header, then a prefix and a body per channel (a body may be empty)
    __shared__ uint16_t s_crc_tab[256];
    __shared__ uint16_t s_crc_zk[3][256];        // CRC of byte x followed by 1, 2, 3 zero bytes (word steps below)
    __shared__ int s_info[FHIP_MAX_CH][4];       // obits, wasted, ch_mode (what verbatim samples are rebuilt from)
    __shared__ uint8_t s_crc8[256];              // crc.c:46-57 (x^8 + x^2 + x + 1)
    static_assert(AT == 64, "the CRC's step from quad to quad is x^(8 * 16 * AT) = crc16_pow_table().v[10]");

    const int lane = threadIdx.x;
    const int nch = P.channels;
    const fhip_subframe_info *fi = info + (size_t)f * nch;
    const int32_t *pcm_frame = pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * nch);
    uint8_t *out = frames + (size_t)f * frame_stride;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    STAMP(0);

    // ---- everything the frame needs from memory but its sections, in flight at once: lane c holds channel c's record,
    // lanes 0 .. 31 / 32 .. 63 every channel's warm-up samples / coefficients ----
    int type = 0, type_code = 0, order = 0, shift = 0, obits = 0, wasted = 0, rn = 0, ch_mode = 0;
    if (lane < nch) {
        const fhip_subframe_info *i = &fi[lane];
        type = i->type; type_code = i->type_code; order = i->order; shift = i->shift; obits = i->obits;
        wasted = i->wasted; rn = i->rice_nbits; ch_mode = i->ch_mode;
    }
    int32_t fld[FHIP_MAX_CH];
#pragma unroll
    for (int c = 0; c < FHIP_MAX_CH; c++) fld[c] = (c < nch) ? ((lane < 32) ? fi[c].warmup[lane] : fi[c].coefs[lane - 32]) : 0;
    const uint32_t number = numbers ? numbers[f] : number_base + (uint32_t)f * number_step;
    // CRC-16 and CRC-8 tables (crc.c:24-57), four entries per lane
#pragma unroll
    for (int q = 0; q < 256 / AT; q++) {
        const int x = lane + AT * q;
        uint32_t c = (uint32_t)x << 8, c8 = (uint32_t)x;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            c = ((c & 0x8000u) ? ((c << 1) ^ 0x8005u) : (c << 1)) & 0xFFFFu;
            c8 = ((c8 & 0x80u) ? ((c8 << 1) ^ 0x07u) : (c8 << 1)) & 0xFFu;
        }
        s_crc_tab[x] = (uint16_t)c;
        s_crc8[x] = (uint8_t)c8;
    }
    for (int i = lane; i < (FHIP_MAX_CH + 1) * ASM_PREFIX_WORDS; i += AT) (&s_bits[0][0])[i] = 0u;
    __syncthreads();                       // (one wave: the LDS writes above are visible to its other lanes)
#pragma unroll
    for (int q = 0; q < 256 / AT; q++) {
        const int x = lane + AT * q;
        uint16_t v = s_crc_tab[x];
#pragma unroll
        for (int z = 0; z < 3; z++) {
            v = (uint16_t)((v << 8) ^ s_crc_tab[v >> 8]);
            s_crc_zk[z][x] = v;
        }
    }
    STAMP(1);

    int bs0 = -1, bs1 = -1;
    {
        const int bs_tab[15] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384};
        for (int q = 0; q < 15; q++) if (n == bs_tab[q]) { bs0 = q; break; }
        if (bs0 < 0) { bs0 = (n <= 256) ? 6 : 7; bs1 = n - 1; }
    }
    // header length: 32 fixed bits, the UTF-8 style number (encode.c:696-716), the optional block
    // size / sample rate fields, CRC-8
    const int nb = (number < 0x80) ? 1 : (ilog2_dev(number) + 4) / 5;
    const int bsb = (bs1 >= 0) ? (bs1 < 256 ? 1 : 2) : 0, srb = (sr_code1 > 0) ? (sr_code1 < 256 ? 1 : 2) : 0;
    const int hbytes = 4 + nb + bsb + srb;
    const int hdr_bits = 8 * hbytes + 8;

    // ---- does the frame take the verbatim fallback? (encode.c:949)  The frame's length, and with it the segment
    // list, follows from lengths alone: lane c has channel c's ----
    const int prec = P.lpc_precision;
    int plen = 0, blen = 0, bkind = 1, bad = 0;
    if (lane < nch) {
        plen = 8 + wasted;                                           // encode.c:871-905
        if (type == FHIP_SUB_CONSTANT) plen += obits;
        else if (type == FHIP_SUB_VERBATIM) { blen = n * obits; bkind = 2; }
        else {
            bad = rn < 0;
            plen += order * obits + ((type == FHIP_SUB_LPC) ? 9 + order * prec : 0);
            blen = max(rn, 0);
        }
    }
    uint32_t incl = wave_incl_scan_u32_dpp((uint32_t)(plen + blen));
    const int verbatim = (__ballot(bad) != 0ull) ||
                         ((((long long)hdr_bits + (long long)__shfl((int)incl, WAVE - 1, WAVE) + 7) >> 3) + 2 > verbatim_size);
    if (verbatim) {
        if (lane < nch) { type = FHIP_SUB_VERBATIM; type_code = FHIP_SUB_VERBATIM; plen = 8 + wasted; blen = n * obits; bkind = 2; }
        incl = wave_incl_scan_u32_dpp((uint32_t)(plen + blen));
    }
    const int total_bits = hdr_bits + __shfl((int)incl, WAVE - 1, WAVE);
    if (lane < nch) {
        const int pos = hdr_bits + (int)incl - plen - blen;
        s_seg[1 + 2 * lane] = AsmSeg{0, lane, plen, pos};
        s_seg[2 + 2 * lane] = AsmSeg{bkind, lane, blen, pos + plen};
        s_info[lane][0] = obits; s_info[lane][1] = wasted; s_info[lane][2] = ch_mode;
    }
    if (lane == 0) s_seg[0] = AsmSeg{0, -1, hdr_bits, 0};
    const int nseg = 1 + 2 * nch;
    {
        // frame header (encode.c:718-764) + CRC-8: a lane per field (the 32 fixed bits, each byte of the UTF-8 style
        // number, the block-size and sample-rate fields), each with the CRC-8 of its bytes moved to the header's end (the
        // CRC is linear: zero bytes in front leave it alone, zero bytes behind are table steps)
        const int ch_mode0 = __shfl(ch_mode, 0, WAVE);
        uint32_t val = 0;
        int bp = 0, fl = 0;                                     // the field's first byte and its bytes
        if (lane == 0) {
            val = (0x7FFCu << 17) | (((uint32_t)P.allow_vbs & 1u) << 16) | (((uint32_t)bs0 & 15u) << 12) |
                  (((uint32_t)sr_code0 & 15u) << 8) |
                  (((uint32_t)(ch_mode0 == FHIP_CH_NOT_STEREO ? nch - 1 : ch_mode0) & 15u) << 4) | (((uint32_t)bps_code & 7u) << 1);
            fl = 4;
        } else if (lane <= nb) {
            const int j = lane - 1, sh = (nb - 1 - j) * 6;
            val = (nb == 1) ? number : (j == 0) ? (((256u - (256u >> nb)) | (number >> sh)) & 0xFFu)
                                                : (0x80u | ((number >> sh) & 0x3Fu));
            bp = 4 + j; fl = 1;
        } else if (lane == 8 && bsb) {
            val = (uint32_t)bs1; bp = 4 + nb; fl = bsb;
        } else if (lane == 9 && srb) {
            val = (uint32_t)sr_code1; bp = 4 + nb + bsb; fl = srb;
        }
        if (fl) or_field(s_bits[0], 8 * bp, 8 * fl, val);
        uint32_t c8 = 0;
#pragma unroll
        for (int z = 3; z >= 0; z--) c8 = s_crc8[c8 ^ ((val >> (8 * z)) & 0xFFu)];
        for (int z = hbytes - bp - fl; z > 0; z--) c8 = s_crc8[c8];
        c8 = wave_xor_u32(fl ? c8 : 0u);
        if (lane == 0) or_field(s_bits[0], 8 * hbytes, 8, c8);
    }
    STAMP(2);

    // ---- per-subframe prefixes (encode.c:871-905, 800-869): a lane per field ----
#pragma unroll
    for (int c = 0; c < FHIP_MAX_CH; c++) {
        if (c < nch) {
            const int type_c = __shfl(type, c, WAVE), order_c = __shfl(order, c, WAVE), obits_c = __shfl(obits, c, WAVE);
            const int wasted_c = __shfl(wasted, c, WAVE), code_c = __shfl(type_code, c, WAVE), shift_c = __shfl(shift, c, WAVE);
            uint32_t *pw = s_bits[c + 1];
            const int pos0 = 8 + wasted_c;
            const uint32_t omask = (obits_c >= 32) ? 0xFFFFFFFFu : ((1u << obits_c) - 1u);
            if (lane == 0) {
                // a zero bit, six bits of type, the wasted-bits flag; then wasted - 1 zeros and a one
                or_field(pw, 0, 8, (((uint32_t)code_c & 63u) << 1) | (wasted_c ? 1u : 0u));
                if (wasted_c) or_field(pw, 8 + wasted_c - 1, 1, 1u);
            }
            if (type_c == FHIP_SUB_CONSTANT) {
                if (lane == 0) or_field(pw, pos0, obits_c, (uint32_t)fld[c] & omask);
            } else if (type_c == FHIP_SUB_FIXED || type_c == FHIP_SUB_LPC) {
                if (lane < order_c) or_field(pw, pos0 + lane * obits_c, obits_c, (uint32_t)fld[c] & omask);
                if (type_c == FHIP_SUB_LPC) {
                    const int pq = pos0 + order_c * obits_c;
                    if (lane == 0) or_field(pw, pq, 9, (((uint32_t)(prec - 1) & 15u) << 5) | ((uint32_t)shift_c & 31u));
                    if (lane >= 32 && lane - 32 < order_c)
                        or_field(pw, pq + 9 + (lane - 32) * prec, prec, (uint32_t)fld[c] & ((1u << prec) - 1u));
                }
            }
        }
    }
    __syncthreads();
    STAMP(3);

    const int body_bytes = (total_bits + 7) >> 3;                 // before the CRC-16
    const int nwords = (body_bytes + 3) >> 2;
    const int nquads = (nwords + 3) >> 2;

    // one output dword from the segments that overlap it (the general case: headers, section borders, verbatim samples)
    auto gen_word = [&](int w) -> uint32_t {
        const int w0 = w * 32, w1 = w0 + 32;
        uint32_t word = 0;
        // the segments lie back to back in ascending order: skip those that end at or before this
        // word (one LDS read each), stop at the first that starts behind it
        int q0 = 0;
        while (q0 < nseg - 1 && s_seg[q0 + 1].dst <= w0) q0++;
        for (int q = q0; q < nseg; q++) {
            const AsmSeg sg = s_seg[q];
            if (sg.dst >= w1) break;
            const int a = max(sg.dst, w0), b = min(sg.dst + sg.nbits, w1);
            if (a >= b) continue;
            const int cnt = b - a;
            const int sp = a - sg.dst;                            // bit offset inside the segment
            uint32_t bitsv;                                       // cnt bits, right aligned
            if (sg.kind == 2) {
                const int c = sg.ch, ob = s_info[c][0], ws = s_info[c][1], cm = s_info[c][2];
                const uint32_t omask = (ob >= 32) ? 0xFFFFFFFFu : ((1u << ob) - 1u);
                int i = sp / ob, offb = sp % ob, got = 0;
                unsigned long long acc = 0;
                while (got < cnt) {
                    const uint32_t v = (uint32_t)asm_sample(pcm_frame, nch, c, i, cm, ws) & omask;
                    const int take = min(ob - offb, cnt - got);
                    const uint32_t piece = (take >= 32) ? v : ((v >> (ob - offb - take)) & ((1u << take) - 1u));
                    acc = (acc << take) | piece;
                    got += take; offb = 0; i++;
                }
                bitsv = (uint32_t)acc;
            } else {
                // 64 source bits that start at the dword holding bit sp
                const int sw = sp >> 5;
                uint32_t hi, lo;
                if (sg.kind == 0) {
                    const uint32_t *src = s_bits[sg.ch + 1];
                    hi = src[sw];
                    lo = (sw + 1 < ASM_PREFIX_WORDS) ? src[sw + 1] : 0u;
                } else {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(
                        rice + ((size_t)f * nch + sg.ch) * (size_t)slot_bytes);
                    hi = __builtin_bswap32(src[sw]);
                    lo = ((long long)(sw + 1) * 4 < slot_bytes) ? __builtin_bswap32(src[sw + 1]) : 0u;
                }
                const unsigned long long x = ((unsigned long long)hi << 32) | lo;
                bitsv = (uint32_t)((x << (sp & 31)) >> (64 - cnt));
            }
            word |= bitsv << (32 - (a - w0) - cnt);
        }
        return word;
    };

    // ---- the frame, four consecutive dwords a lane and step; its CRC-16 on the way ----------------
    // Most quads lie inside one residual section: five source dwords, four funnel shifts, one 16-byte store; the loads
    // of ASM_QB quads are issued before the first is used.
    // CRC-16 (crc.c:59-94): CRC(A || B) = CRC(A) x^(8 |B|) + CRC(B) over GF(2).  A lane runs the table step over its
    // quad's four dwords, carries the CRC of its own quads -- 16 AT bytes apart: one constant product a quad -- moves it
    // to the end of the frame's last quad (lane t's last quad is the t-th from the end), and the 64 values are XORed;
    // the zero bytes that fill the last quad are taken off again by x^-8 (x^32767 = 1).  (Round 3 read the finished
    // frame back from memory with one wave of four, a load the next step waited for at each dword, and produced the
    // frame a dword at a time with 64-bit positions.)
    const bool st16 = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const uint32_t *rice0 = reinterpret_cast<const uint32_t *>(rice + (size_t)f * nch * (size_t)slot_bytes);
    const int slot_dw = (int)(slot_bytes >> 2);
    uint32_t crc = 0;
    const int last = nquads - 1 - lane;                           // lane t's quads are last - AT i, in ascending order
    int q0 = 0;                                                   // (the segment of a quad's first bit: quads ascend)
    for (int base = last & (AT - 1); base <= last; base += AT * ASM_QB) {
        uint32_t raw[ASM_QB][5];
        int dq[ASM_QB], segq[ASM_QB];
        bool fastq[ASM_QB];
#pragma unroll
        for (int j = 0; j < ASM_QB; j++) {
            const int qd = base + AT * j;
            const int b0 = qd * 128;
            while (q0 < nseg - 1 && s_seg[q0 + 1].dst <= b0) q0++;
            const AsmSeg sg = s_seg[q0];
            const bool sec = qd <= last && sg.kind == 1;
            // 128 bits of the section from its bit b0 - dst on (a quad that starts elsewhere, or none at all, loads the
            // frame's first dwords: no branch around the loads; dwords past the slot are never inside the section)
            const int d = sec ? b0 - sg.dst : 0;
            const uint32_t *src = rice0 + (sec ? (size_t)sg.ch * (size_t)slot_dw : (size_t)0);
            dq[j] = d;
            segq[j] = q0;
            fastq[j] = sec && b0 + 128 <= sg.dst + sg.nbits;
#pragma unroll
            for (int k = 0; k < 5; k++) raw[j][k] = src[min((d >> 5) + k, slot_dw - 1)];
        }
#pragma unroll
        for (int j = 0; j < ASM_QB; j++) {
            const int qd = base + AT * j;
            if (qd <= last) {
                uint32_t wd[4];
                if (fastq[j]) {
                    const int sh = dq[j] & 31;
                    uint32_t sv[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) sv[k] = __builtin_bswap32(raw[j][k]);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t fs = __builtin_amdgcn_alignbit(sv[k], sv[k + 1], (32 - sh) & 31);
                        wd[k] = sh ? fs : sv[k];
                    }
                } else {
                    // a quad with a border in it: every segment that overlaps it gives its 128-bit window (the first one's
                    // is loaded already), cut to the bits that are its own
                    const int b0 = qd * 128;
                    wd[0] = wd[1] = wd[2] = wd[3] = 0u;
                    bool samples = false;
                    for (int i = segq[j]; i < nseg; i++) {
                        const AsmSeg sg = s_seg[i];
                        if (sg.dst >= b0 + 128) break;
                        const int lo = max(sg.dst, b0) - b0, hi = min(sg.dst + sg.nbits, b0 + 128) - b0;
                        if (lo >= hi) continue;
                        if (sg.kind == 2) { samples = true; break; }
                        const int d = b0 - sg.dst;                 // (negative: the segment starts inside the quad)
                        const int sw = d >> 5, sh = d & 31;
                        uint32_t sv[5];
                        if (sg.kind == 0) {
                            const uint32_t *src = s_bits[sg.ch + 1];
#pragma unroll
                            for (int k = 0; k < 5; k++) sv[k] = (sw + k >= 0 && sw + k < ASM_PREFIX_WORDS) ? src[sw + k] : 0u;
                        } else if (i == segq[j]) {
#pragma unroll
                            for (int k = 0; k < 5; k++) sv[k] = __builtin_bswap32(raw[j][k]);
                        } else {
                            const uint32_t *src = rice0 + (size_t)sg.ch * (size_t)slot_dw;
#pragma unroll
                            for (int k = 0; k < 5; k++) sv[k] = (sw + k >= 0) ? __builtin_bswap32(src[min(sw + k, slot_dw - 1)]) : 0u;
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t fs = __builtin_amdgcn_alignbit(sv[k], sv[k + 1], (32 - sh) & 31);
                            const int l = lo - 32 * k, h = hi - 32 * k;     // the segment's bits of this dword: [l, h), bit 0 on top
                            const uint32_t m = ((l <= 0) ? 0xFFFFFFFFu : (l >= 32) ? 0u : (0xFFFFFFFFu >> l)) &
                                               ((h >= 32) ? 0xFFFFFFFFu : (h <= 0) ? 0u : ~(0xFFFFFFFFu >> h));
                            wd[k] |= (sh ? fs : sv[k]) & m;
                        }
                    }
                    if (samples) {
                        // verbatim samples (a frame that took the fallback, or a subframe K3 left verbatim): dword by dword
#pragma unroll 1
                        for (int k = 0; k < 4; k++) {
                            const int w = 4 * qd + k;
                            const uint32_t v = (w < nwords) ? gen_word(w) : 0u;
                            wd[0] = (k == 0) ? v : wd[0]; wd[1] = (k == 1) ? v : wd[1];
                            wd[2] = (k == 2) ? v : wd[2]; wd[3] = (k == 3) ? v : wd[3];
                        }
                    }
                }
                if (st16 && 4 * qd + 4 <= nwords) {
                    *reinterpret_cast<uint4 *>(out32 + 4 * qd) = make_uint4(__builtin_bswap32(wd[0]), __builtin_bswap32(wd[1]),
                                                                            __builtin_bswap32(wd[2]), __builtin_bswap32(wd[3]));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (4 * qd + k < nwords) out32[4 * qd + k] = __builtin_bswap32(wd[k]);
                }
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t x0 = ((c >> 8) ^ (wd[k] >> 24)) & 0xFFu, x1 = (c ^ (wd[k] >> 16)) & 0xFFu;
                    c = (uint32_t)(s_crc_zk[2][x0] ^ s_crc_zk[1][x1] ^ s_crc_zk[0][(wd[k] >> 8) & 0xFFu] ^ s_crc_tab[wd[k] & 0xFFu]);
                }
                crc = crc16_mul_const<crc16_pow_table().v[10]>(crc) ^ c;
            }
        }
    }
    STAMP(4);
    // behind the `lane` quads that follow this lane's last one
    crc = crc16_shift_bytes<9, 4>(crc, 16 * lane);
    crc = wave_xor_u32(crc);
    __threadfence_block();                 // the dword the two CRC bytes lie in may be one this wave stored above
    if (lane == 0) {
        const uint32_t c = crc16_unshift_bytes(crc, 16 * nquads - body_bytes);
        out[body_bytes] = (uint8_t)(c >> 8);
        out[body_bytes + 1] = (uint8_t)c;
        frame_bytes[f] = body_bytes + 2;
    }
    STAMP(6);
}


// ---------------------------------------------------------------------------
// K-vbs  k_vbs_split -- vbs.c:36-83 split_frame_v1
// ---------------------------------------------------------------------------
// One workgroup per block: eight sections of n/8 sample-frames, for each the
// sum over channels of |x[j] - 2x[j-1] + x[j-2]| on the raw interleaved input
// (int32 wrap, then abs), divided by the channel count, plus one; neighbours
// are merged unless the score changes by more than 25 % -- evaluated with the
// reference's int abs() and 32-bit multiply.
__global__ __launch_bounds__(NT)
void k_vbs_split(const int32_t *__restrict__ pcm, int nblocks, int block_size, int nch,
                 int32_t *__restrict__ nframes_out, int32_t *__restrict__ sizes_out)
{
    __shared__ long long s_score[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    const int n = block_size / 8;
    const int32_t *base = pcm + (size_t)b * block_size * nch;
    for (int sec = wv; sec < 8; sec += 4) {
        const int32_t *sp = base + (size_t)sec * n * nch;
        long long acc = 0;
        // element e of the section = (j, ch) interleaved; rows j >= 2 only
        const int total = (n - 2) * nch;
        if (nch == 2 && (n & 1) == 0) {
            // stereo (sections of an even length: 16-byte aligned): four elements per lane and step from two aligned 16-byte loads (rows j-2 .. j+1 of
            // both channels); 2 n - 4 elements are whole groups of four
            for (int e = 4 * lane; e < total; e += 4 * WAVE) {
                const int4 lo = *reinterpret_cast<const int4 *>(sp + e);          // elements idx-4 .. idx-1
                const int4 hi = *reinterpret_cast<const int4 *>(sp + e + 4);      // idx .. idx+3
                const uint32_t w[8] = {(uint32_t)lo.x, (uint32_t)lo.y, (uint32_t)lo.z, (uint32_t)lo.w,
                                       (uint32_t)hi.x, (uint32_t)hi.y, (uint32_t)hi.z, (uint32_t)hi.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t d = (int32_t)(w[4 + k] - 2u * w[2 + k] + w[k]);
                    acc += (long long)wrap_abs(d);
                }
            }
        } else {
            for (int e = lane; e < total; e += WAVE) {
                const int idx = e + 2 * nch;
                const uint32_t x0 = (uint32_t)sp[idx], x1 = (uint32_t)sp[idx - nch], x2 = (uint32_t)sp[idx - 2 * nch];
                const int32_t d = (int32_t)(x0 - 2u * x1 + x2);
                acc += (long long)wrap_abs(d);
            }
        }
        acc = (long long)wave_sum_u64((unsigned long long)acc);
        if (lane == 0) s_score[sec] = acc / nch + 1;
    }
    __syncthreads();
    if (tid == 0) {
        int sizes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nf = 0;
        for (int p = 0; p < 8; p++) {
            bool cut = (p == 0);
            if (p > 0) {
                int32_t diff = (int32_t)(uint32_t)(unsigned long long)(s_score[p - 1] - s_score[p]);
                diff = wrap_abs(diff);
                const int32_t scaled = (int32_t)((uint32_t)diff * 200u);
                cut = ((long long)scaled / s_score[p - 1]) > 50;
            }
            if (cut) nf++;
            sizes[nf - 1] += n;
        }
        nframes_out[b] = nf;
        for (int p = 0; p < 8; p++) sizes_out[(size_t)b * 8 + p] = sizes[p];
    }
}

}  // namespace

hipError_t launch_vbs_split(hipStream_t st, const int32_t *pcm, int nblocks, int block_size,
                            int nch, int32_t *nframes_out, int32_t *sizes_out)
{
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vbs_split, dim3(nblocks), dim3(NT), 0, st, pcm, nblocks, block_size, nch,
                       nframes_out, sizes_out);
    return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t st, const fhip_params &p, const int32_t *pcm, int nframes,
                           int n, const fhip_subframe_info *info, const uint8_t *rice,
                           int64_t slot_bytes, uint8_t *frames, int64_t frame_stride,
                           int32_t *frame_bytes, uint32_t number_base, uint32_t number_step,
                           const uint32_t *numbers, const long long *frame_src, const int32_t *dev_frames)
{
    if (nframes == 0) return hipSuccess;
    // sample-rate / bit-depth codes of flake_encode_init() (encode.c:400-438)
    static const int sr_table[16] = {0, 0, 0, 0, 8000, 16000, 22050, 24000, 32000, 44100, 48000,
                                     96000, 0, 0, 0, 0};
    static const int bd_table[8] = {0, 8, 12, 0, 16, 20, 24, 0};
    int sr0 = 0, sr1 = 0, bpsc = 0;
    for (int i = 4; i < 12; i++) if (p.sample_rate == sr_table[i]) { sr0 = i; break; }
    if (!sr0) {
        const int sr = p.sample_rate;
        if (sr % 1000 == 0 && sr <= 255000) { sr0 = 12; sr1 = sr / 1000; }
        else if (sr % 10 == 0 && sr <= 655350) { sr0 = 14; sr1 = sr / 10; }
        else if (sr < 65535) { sr0 = 13; sr1 = sr; }
    }
    for (int i = 1; i < 8; i++) if (p.bits_per_sample == bd_table[i]) { bpsc = i; break; }
    const int bps = p.bits_per_sample;
    const int vsize = (p.channels == 2) ? 16 + ((n * (bps + bps + 1) + 7) >> 3)
                                        : 16 + ((n * p.channels * bps + 7) >> 3);
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(AT), 0, st, p, n, pcm, info, rice,
                       (long long)slot_bytes, frames, (long long)frame_stride, frame_bytes,
                       number_base, number_step, numbers, sr0, sr1, bpsc, vsize, frame_src, dev_frames, MultiBin{});
    return hipGetLastError();
}

hipError_t launch_assemble_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *pcm,
                                const fhip_subframe_info *info, const uint8_t *rice, uint8_t *frames,
                                int32_t *frame_bytes, const uint32_t *numbers, const long long *frame_src)
{
    if (mb.nbins < 1 || !numbers || !frame_src) return hipErrorInvalidValue;
    const int slots = mb.wg0[mb.nbins];
    if (slots == 0) return hipSuccess;
    static const int sr_table[16] = {0, 0, 0, 0, 8000, 16000, 22050, 24000, 32000, 44100, 48000,
                                     96000, 0, 0, 0, 0};
    static const int bd_table[8] = {0, 8, 12, 0, 16, 20, 24, 0};
    int sr0 = 0, sr1 = 0, bpsc = 0;
    for (int i = 4; i < 12; i++) if (p.sample_rate == sr_table[i]) { sr0 = i; break; }
    if (!sr0) {
        const int sr = p.sample_rate;
        if (sr % 1000 == 0 && sr <= 255000) { sr0 = 12; sr1 = sr / 1000; }
        else if (sr % 10 == 0 && sr <= 655350) { sr0 = 14; sr1 = sr / 10; }
        else if (sr < 65535) { sr0 = 13; sr1 = sr; }
    }
    for (int i = 1; i < 8; i++) if (p.bits_per_sample == bd_table[i]) { bpsc = i; break; }
    hipLaunchKernelGGL(k_assemble, dim3(slots), dim3(AT), 0, st, p, 0, pcm, info, rice, 0ll, frames, 0ll,
                       frame_bytes, 0u, 0u, numbers, sr0, sr1, bpsc, 0, frame_src, (const int32_t *)nullptr, mb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K4-P  the batch's frames back to back (what flake_encode_frame's callers write to the file)
// ---------------------------------------------------------------------------
namespace {

constexpr int SCAN_NT = 1024;

// offsets[f] = bytes of frames 0 .. f-1 (exclusive scan of frame_bytes, negatives count 0);
// offsets[nframes] = the batch's stream length.  One workgroup.
__global__ __launch_bounds__(SCAN_NT)
void k_frame_offsets(const int32_t *__restrict__ fbytes, int nframes, long long *__restrict__ offsets)
{
    __shared__ long long s_part[SCAN_NT];
    const int tid = threadIdx.x;
    const int per = (nframes + SCAN_NT - 1) / SCAN_NT;
    const int f0 = tid * per, f1 = min(f0 + per, nframes);
    long long sum = 0;
    for (int f = f0; f < f1; f++) sum += max(fbytes[f], 0);
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_NT; off <<= 1) {          // Hillis-Steele inclusive scan
        const long long v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    long long run = s_part[tid] - sum;
    for (int f = f0; f < f1; f++) { offsets[f] = run; run += max(fbytes[f], 0); }
    if (tid == SCAN_NT - 1) offsets[nframes] = s_part[tid];
}

// One workgroup per frame: its bytes from the (4-byte aligned) slot to byte offset
// offsets[f] of the packed stream.  Destination dwords are formed from two source dwords
// (v_alignbyte); up to three bytes at either end go one at a time.
__global__ __launch_bounds__(NT)
void k_pack_frames(const uint8_t *__restrict__ frames, long long stride, const int32_t *__restrict__ fbytes,
                   const long long *__restrict__ offsets, uint8_t *__restrict__ packed)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const int len = max(fbytes[f], 0);
    const uint8_t *src = frames + (size_t)f * (size_t)stride;
    uint8_t *dst = packed + offsets[f];
    const int head = min((int)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3), len);
    if (tid < head) dst[tid] = src[tid];
    const int ndw = (len - head) >> 2;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + head);
    const int sh = head & 3;                               // source byte phase of every destination dword
    for (int j = tid; j < ndw; j += NT) {
        const uint32_t lo = s32[j + (head >> 2)];
        const uint32_t hi = sh ? s32[j + (head >> 2) + 1] : 0u;   // inside the slot: stride has 8 bytes of slack
        d32[j] = __builtin_amdgcn_alignbyte(hi, lo, sh);
    }
    const int done = head + 4 * ndw;
    if (tid < len - done) dst[done + tid] = src[done + tid];
}

// ---------------------------------------------------------------------------
// Ragged (VBS) batches, device-resident  (vbs.c:85-119 + encode.c:996-1004, batched)
// ---------------------------------------------------------------------------
// split_frame_v1's verdicts stay on the device.  A piece is k eighths of its block, k = 1 .. 8, so
// there are eight BINS of equal piece length; bin k-1 owns a fixed range of frame slots sized for
// the most pieces it can get (floor(8 / k) per block: VbsBins, host constants), and only the
// number of pieces that really fell into it -- cnt[] -- lives on the device.  The path's kernels
// run once per bin on a grid sized for the bin's capacity and read their count from cnt[]
// (dev_count); the frames of all bins are then packed in stream order.
//
// k_vbs_plan (one workgroup): from nframes[b] / sizes[b][8] of k_vbs_split
//   cnt[0..7]   frames per bin          cnt[8..15]  subframes per bin       cnt[16]  frames in all
//   order[i]    slot of the stream's i-th frame (stream order = block order, pieces in order)
//   per slot:   frame_src (offset of the piece's PCM in int32 units), src_off (byte offset of the
//               slot's frame in frames[]), numbers (its first sample: encode.c:969-975, allow_vbs)
//   first[b]    stream index of block b's first frame (first[nblocks] = frames in all)
// A block the splitter left whole (nframes <= 1) is one piece of eight eighths (vbs.c:100,
// encode.c:1001).
constexpr int PLAN_NT = 1024;

__global__ __launch_bounds__(PLAN_NT)
void k_vbs_plan(const int32_t *__restrict__ nfr, const int32_t *__restrict__ sizes, int nblocks,
                int block_size, int nch, uint32_t first_number, VbsBins bins,
                int32_t *__restrict__ cnt, int32_t *__restrict__ order, long long *__restrict__ frame_src,
                long long *__restrict__ src_off, uint32_t *__restrict__ numbers, int32_t *__restrict__ first)
{
    // Round 4: a thread owns one block of every 1024 (coalesced reads of the splitter's verdicts, the running totals
    // carried from chunk to chunk) -- a thread that owned nblocks / 1024 consecutive blocks walked them by dependent
    // loads twice over: 123 us for 8192 blocks.
    __shared__ int32_t s_wtot[9][PLAN_NT / 64 + 1];
    __shared__ int s_slot0[8];
    __shared__ long long s_froff[8], s_stride[8];       // (the bins' constants by a piece's bin: an LDS read, not a select chain)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int eighth = block_size / 8;
    const float inv_eighth = 1.0f / (float)eighth;      // a piece is 1 .. 8 whole eighths: len / eighth by one rounding
    if (tid < 8) { s_slot0[tid] = bins.slot0[tid]; s_froff[tid] = bins.fr_off[tid]; s_stride[tid] = bins.stride[tid]; }
    int base[9];
#pragma unroll
    for (int k = 0; k < 9; k++) base[k] = 0;
    for (int b0 = 0; b0 < nblocks; b0 += PLAN_NT) {
        const int b = b0 + tid;
        const bool on = b < nblocks;
        int f = 0;
        int len[8];
#pragma unroll
        for (int q = 0; q < 8; q++) len[q] = 0;
        if (on) {
            f = nfr[b];
            const int4 s0 = *reinterpret_cast<const int4 *>(sizes + (size_t)b * 8);
            const int4 s1 = *reinterpret_cast<const int4 *>(sizes + (size_t)b * 8 + 4);
            len[0] = s0.x; len[1] = s0.y; len[2] = s0.z; len[3] = s0.w;
            len[4] = s1.x; len[5] = s1.y; len[6] = s1.z; len[7] = s1.w;
            if (f <= 1) { f = 1; len[0] = block_size; }          // the splitter left the block whole
        }
        int bin[8], c[9];
#pragma unroll
        for (int k = 0; k < 9; k++) c[k] = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            bin[q] = (q < f) ? (int)((float)len[q] * inv_eighth + 0.5f) - 1 : -1;
#pragma unroll
            for (int z = 0; z < 8; z++) c[z] += (z == bin[q]);
        }
        c[8] = f;
        // nine exclusive scans over the 1024 threads: inside a wave by DPP, the sixteen wave totals by the
        // first lanes of wave 0
        int incl[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            incl[k] = (int)wave_incl_scan_u32_dpp((uint32_t)c[k]);
            if (lane == 63) s_wtot[k][wv] = incl[k];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int t = (lane < PLAN_NT / 64) ? s_wtot[k][lane] : 0;
                const int sc = (int)wave_incl_scan_u32_dpp((uint32_t)t);
                if (lane < PLAN_NT / 64) s_wtot[k][lane] = sc - t;     // exclusive prefix of the wave
                if (lane == PLAN_NT / 64 - 1) s_wtot[k][PLAN_NT / 64] = sc;   // the chunk's total
            }
        }
        __syncthreads();
        int run[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            run[k] = base[k] + s_wtot[k][wv] + incl[k] - c[k];
            base[k] += s_wtot[k][PLAN_NT / 64];
        }
        if (on) {
            first[b] = run[8];
            long long pos = (long long)b * block_size;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q < f) {
                    const int k = bin[q];
                    int j = 0;
#pragma unroll
                    for (int z = 0; z < 8; z++) if (z == k) { j = run[z]; run[z]++; }
                    const int slot = s_slot0[k] + j;
                    order[run[8]++] = slot;
                    frame_src[slot] = pos * nch;
                    src_off[slot] = s_froff[k] + (long long)j * s_stride[k];
                    numbers[slot] = first_number + (uint32_t)pos;
                    pos += len[q];
                }
            }
        }
        __syncthreads();                     // s_wtot is written again by the next chunk
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) { cnt[k] = base[k]; cnt[8 + k] = base[k] * nch; }
        cnt[16] = base[8];
        first[nblocks] = base[8];
    }
}

// Round 4: chunks of 4096 frames, four consecutive frames per thread (one 16-byte read of order[], four gathers in
// flight), wave scans and one barrier per chunk -- a thread that walked nframes / 1024 consecutive frames by dependent
// loads, twice, and a Hillis-Steele scan of twenty barriers took 70-92 us for the frames of 8192 blocks.
__global__ __launch_bounds__(SCAN_NT)
void k_frame_offsets_perm(const int32_t *__restrict__ fbytes, const int32_t *__restrict__ order,
                          const int32_t *__restrict__ dev_frames, long long *__restrict__ offsets,
                          long long cap, long long *__restrict__ totals)
{
    __shared__ long long s_w[2][SCAN_NT / 64];       // the wave totals of a chunk, by the chunk's parity
    __shared__ int s_max, s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nframes = dev_count(dev_frames, 0);
    const bool vec = (reinterpret_cast<uintptr_t>(order) & 15) == 0;
    long long base = 0;
    int mx = 0, bad = 0, par = 0;
    if (tid == 0) { s_max = 0; s_bad = 0; }
    for (int c0 = 0; c0 < nframes; c0 += 4 * SCAN_NT, par ^= 1) {
        const int f = c0 + 4 * tid;
        int slot[4], b[4];
        if (vec && f + 4 <= nframes) {
            const int4 o = *reinterpret_cast<const int4 *>(order + f);
            slot[0] = o.x; slot[1] = o.y; slot[2] = o.z; slot[3] = o.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) slot[q] = (f + q < nframes) ? order[f + q] : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) b[q] = (slot[q] >= 0) ? fbytes[slot[q]] : 1;
        long long sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bad |= (b[q] <= 0);                      // a frame K4 did not produce: the stream would be short of it
            b[q] = (slot[q] >= 0) ? max(b[q], 0) : 0;
            mx = max(mx, b[q]);
            sum += b[q];
        }
        const long long incl = (long long)wave_incl_scan_u64((unsigned long long)sum, lane);
        if (lane == 63) s_w[par][wv] = incl;
        __syncthreads();
        long long pre = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < SCAN_NT / 64; w++) {
            const long long t = s_w[par][w];
            tot += t;
            pre += (w < wv) ? t : 0;
        }
        long long run = base + pre + incl - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (f + q < nframes) offsets[f + q] = run;
            run += b[q];
        }
        base += tot;
    }
    __syncthreads();                                 // s_max / s_bad are zero (a batch without frames runs no chunk)
    if (mx > 0) atomicMax(&s_max, mx);
    if (bad) atomicOr(&s_bad, 1);
    __syncthreads();
    if (tid == 0) {
        offsets[nframes] = base;
        // totals: frames, bytes, largest frame, flags: 1 = the stream does not fit `cap`, 2 = some frame of the
        // stream was not encoded (frame_bytes <= 0: its bytes are missing from the packed stream)
        totals[0] = nframes;
        totals[1] = base;
        totals[2] = s_max;                     // encode.c:967
        totals[3] = ((base > cap) ? 1 : 0) | (s_bad ? 2 : 0);
    }
}

__global__ __launch_bounds__(NT)
void k_pack_frames_perm(const uint8_t *__restrict__ frames, const long long *__restrict__ src_off,
                        const int32_t *__restrict__ fbytes, const int32_t *__restrict__ order,
                        const long long *__restrict__ offsets, uint8_t *__restrict__ packed,
                        const int32_t *__restrict__ dev_frames, long long cap,
                        int32_t *__restrict__ stream_bytes, long long *__restrict__ totals)
{
    const int tid = threadIdx.x;
    const int total = dev_count(dev_frames, 0);
    for (int f = blockIdx.x; f < total; f += gridDim.x) {         // (a grid of the most frames there can be: mostly empty workgroups)
    const int slot = order[f];
    const int len = max(fbytes[slot], 0);
    if (tid == 0 && stream_bytes) stream_bytes[f] = fbytes[slot];   // the frame sizes in stream order
    if (offsets[f] + len > cap) continue;                           // the caller's buffer ends here (totals[3])
    const uint8_t *src = frames + src_off[slot];                   // 4-byte aligned
    uint8_t *dst = packed + offsets[f];
    const int head = min((int)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3), len);
    if (tid < head) dst[tid] = src[tid];
    const int ndw = (len - head) >> 2;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + head);
    const int sh = head & 3;
    for (int j = tid; j < ndw; j += NT) {
        const uint32_t lo = s32[j + (head >> 2)];
        const uint32_t hi = sh ? s32[j + (head >> 2) + 1] : 0u;
        d32[j] = __builtin_amdgcn_alignbyte(hi, lo, sh);
    }
    const int done = head + 4 * ndw;
    if (tid < len - done) dst[done + tid] = src[done + tid];
    }
}

// bytes and frames of every block (what flake_encode_frame returns for it, vbs.c:104-116)
__global__ __launch_bounds__(NT)
void k_vbs_block_bytes(const int32_t *__restrict__ first, const long long *__restrict__ offsets, int nblocks,
                       int32_t *__restrict__ block_bytes, int32_t *__restrict__ block_frames)
{
    const int b = blockIdx.x * NT + threadIdx.x;
    if (b >= nblocks) return;
    const int i0 = first[b], i1 = first[b + 1];
    if (block_bytes) block_bytes[b] = (int32_t)(offsets[i1] - offsets[i0]);
    if (block_frames) block_frames[b] = i1 - i0;
}

}  // namespace

hipError_t launch_vbs_plan(hipStream_t st, const int32_t *nfr, const int32_t *sizes, int nblocks,
                           int block_size, int nch, uint32_t first_number, const VbsBins &bins,
                           int32_t *cnt, int32_t *order, long long *frame_src, long long *src_off,
                           uint32_t *numbers, int32_t *first)
{
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vbs_plan, dim3(1), dim3(PLAN_NT), 0, st, nfr, sizes, nblocks, block_size, nch,
                       first_number, bins, cnt, order, frame_src, src_off, numbers, first);
    return hipGetLastError();
}

hipError_t launch_pack_frames_perm(hipStream_t st, const uint8_t *frames, const long long *src_off,
                                   const int32_t *frame_bytes, const int32_t *order, int max_frames,
                                   const int32_t *dev_frames, long long *offsets, uint8_t *packed,
                                   long long cap, int32_t *stream_bytes, long long *totals)
{
    if (max_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frame_offsets_perm, dim3(1), dim3(SCAN_NT), 0, st, frame_bytes, order, dev_frames,
                       offsets, cap, totals);
    hipLaunchKernelGGL(k_pack_frames_perm, dim3(max_frames), dim3(NT), 0, st, frames, src_off, frame_bytes,
                       order, offsets, packed, dev_frames, cap, stream_bytes, totals);
    return hipGetLastError();
}

hipError_t launch_vbs_block_bytes(hipStream_t st, const int32_t *first, const long long *offsets, int nblocks,
                                  int32_t *block_bytes, int32_t *block_frames)
{
    if (nblocks == 0 || (!block_bytes && !block_frames)) return hipSuccess;
    hipLaunchKernelGGL(k_vbs_block_bytes, dim3((nblocks + NT - 1) / NT), dim3(NT), 0, st, first, offsets,
                       nblocks, block_bytes, block_frames);
    return hipGetLastError();
}

hipError_t launch_pack_frames(hipStream_t st, const uint8_t *frames, int64_t frame_stride,
                              const int32_t *frame_bytes, int nframes, long long *offsets,
                              uint8_t *packed)
{
    if (nframes == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frame_offsets, dim3(1), dim3(SCAN_NT), 0, st, frame_bytes, nframes, offsets);
    hipLaunchKernelGGL(k_pack_frames, dim3(nframes), dim3(NT), 0, st, frames, (long long)frame_stride,
                       frame_bytes, offsets, packed);
    return hipGetLastError();
}

}  // namespace fhip
