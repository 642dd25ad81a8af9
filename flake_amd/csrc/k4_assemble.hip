// k4_assemble.hip -- K4: whole FLAC frames on the device (encode.c:718-764, :800-917,
// :949-964) and the VBS splitter (vbs.c:36-83).
#include "device_util.h"

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

struct AsmSeg { int kind; int ch; long long nbits; long long dst; };   // kind: 0 LDS bytes, 1 rice slot, 2 verbatim samples

struct MiniSink {                          // MSB-first writer into LDS bytes (serial, one thread)
    // Fields gather in a 64-bit register and leave a byte at a time by plain byte stores (round 2 set
    // every bit by a read-modify-write of its LDS byte: ~400 dependent LDS round trips per subframe
    // prefix, the latency a frame's workgroup waited for).  No recursion: the compiler keeps it inline.
    uint8_t *buf; int nbits;
    unsigned long long acc = 0; int nacc = 0;
    __device__ __forceinline__ void emit(int nb, uint32_t v)      // 1 <= nb <= 32
    {
        const uint32_t m = (nb == 32) ? 0xFFFFFFFFu : ((1u << nb) - 1u);
        acc = (acc << nb) | (unsigned long long)(v & m);
        nacc += nb;
#pragma unroll
        for (int z = 0; z < 5; z++) {
            if (nacc >= 8) {
                nacc -= 8;
                buf[nbits >> 3] = (uint8_t)(acc >> nacc);
                nbits += 8;
            }
        }
        // the pending bits, left-aligned, as the (partial) last byte: readers see a complete byte image
        buf[nbits >> 3] = (uint8_t)((acc << (8 - nacc)) & 0xFFu);
    }
    __device__ __forceinline__ void put(int nb, uint32_t v)
    {
        if (nb <= 0) return;
        if (nb > 32) { emit(nb - 32, 0u); nb = 32; }               // fields wider than 32 bits are zero-extended
        emit(nb, v);
    }
    __device__ __forceinline__ int bits() const { return nbits + nacc; }
};

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

__global__ __launch_bounds__(NT)
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
        // every bin of a ragged batch in one launch (kernels.h: MultiBin; one workgroup per frame
        // slot: wg0 = unit0): info / frame_bytes / numbers / frame_src are the handle's whole
        // slot-indexed arrays, the sections and the frames lie bin by bin
        const int k = find_bin(mb, f);
        const int local = f - mb.unit0[k];
        if (local >= __builtin_amdgcn_readfirstlane(mb.cnt[mb.cnt_ix[k]])) return;
        n = mb.n[k];
        verbatim_size = mb.vsize[k];
        slot_bytes = mb.slot[k];
        frame_stride = mb.stride[k];
        rice += mb.bits_off[k] - (long long)mb.unit0[k] * P.channels * slot_bytes;     // indexed by the global slot below
        frames += mb.fr_off[k] - (long long)mb.unit0[k] * frame_stride;
    }
    __shared__ uint8_t s_hdr[32];
    __shared__ uint8_t s_prefix[FHIP_MAX_CH][ASM_PREFIX_BYTES];
    __shared__ AsmSeg s_seg[ASM_MAX_SEG];
    __shared__ int s_nseg, s_verbatim, s_hdr_bits;
    __shared__ long long s_total_bits;
    __shared__ uint16_t s_crc_tab[256];
    __shared__ uint16_t s_crc_zk[3][256];        // CRC of byte x followed by 1, 2, 3 zero bytes (word steps below)
    __shared__ int s_info[FHIP_MAX_CH][8];       // type, type_code, order, shift, obits, wasted, rice_nbits, ch_mode

    const int tid = threadIdx.x;
    const int nch = P.channels;
    const fhip_subframe_info *fi = info + (size_t)f * nch;
    const int32_t *pcm_frame = pcm + (frame_src ? (size_t)frame_src[f] : (size_t)f * n * nch);
    uint8_t *out = frames + (size_t)f * frame_stride;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);

    // CRC-16 table (crc.c:24-44), one entry per thread
    {
        uint16_t c = (uint16_t)(tid << 8);
#pragma unroll
        for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
        s_crc_tab[tid] = c;
    }
    if (tid < nch) {
        const fhip_subframe_info *i = &fi[tid];
        s_info[tid][0] = i->type; s_info[tid][1] = i->type_code; s_info[tid][2] = i->order;
        s_info[tid][3] = i->shift; s_info[tid][4] = i->obits; s_info[tid][5] = i->wasted;
        s_info[tid][6] = i->rice_nbits; s_info[tid][7] = i->ch_mode;
    }
    __syncthreads();
    {
        uint16_t v = s_crc_tab[tid];
#pragma unroll
        for (int z = 0; z < 3; z++) {
            v = (uint16_t)((v << 8) ^ s_crc_tab[v >> 8]);
            s_crc_zk[z][tid] = v;                 // read first behind later barriers
        }
    }

    // ---- does the frame take the verbatim fallback? (encode.c:949) -----------
    // The frame's length follows from lengths alone; the header's bytes (below, thread 64) and the
    // subframe prefixes (threads 0 .. channels-1) are then written side by side, in different waves.
    int bs0 = -1, bs1 = -1;
    {
        const int bs_tab[15] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384};
        for (int q = 0; q < 15; q++) if (n == bs_tab[q]) { bs0 = q; break; }
        if (bs0 < 0) { bs0 = (n <= 256) ? 6 : 7; bs1 = n - 1; }
    }
    const uint32_t number = numbers ? numbers[f] : number_base + (uint32_t)f * number_step;
    if (tid == 0) {
        // header length: 32 fixed bits, the UTF-8 style number (encode.c:696-716), the optional block
        // size / sample rate fields, CRC-8
        const int nbytes = (number < 0x80) ? 1 : (ilog2_dev(number) + 4) / 5;
        const int hdr_bits = 32 + 8 * nbytes + (bs1 >= 0 ? (bs1 < 256 ? 8 : 16) : 0) +
                             (sr_code1 > 0 ? (sr_code1 < 256 ? 8 : 16) : 0) + 8;
        s_hdr_bits = hdr_bits;
        long long bits = hdr_bits;
        int verb = 0;
        for (int c = 0; c < nch; c++) {
            const int type = s_info[c][0], order = s_info[c][2], obits = s_info[c][4];
            const int wasted = s_info[c][5], rn = s_info[c][6];
            bits += 8 + (wasted ? wasted : 0);
            if (type == FHIP_SUB_CONSTANT) bits += obits;
            else if (type == FHIP_SUB_VERBATIM) bits += (long long)n * obits;
            else {
                if (rn < 0) verb = 1;
                bits += (long long)order * obits + rn;
                if (type == FHIP_SUB_LPC) bits += 9 + order * P.lpc_precision;
            }
        }
        const long long bytes = ((bits + 7) >> 3) + 2;
        if (bytes > verbatim_size) verb = 1;
        s_verbatim = verb;
    }
    __syncthreads();
    const int verbatim = s_verbatim;

    if (tid == WAVE) {
        // frame header (encode.c:718-764) + CRC-8
        MiniSink hs{s_hdr, 0};
        const int ch_mode = s_info[0][7];
        hs.put(15, 0x7FFC);
        hs.put(1, (uint32_t)P.allow_vbs);
        hs.put(4, (uint32_t)bs0);
        hs.put(4, (uint32_t)sr_code0);
        hs.put(4, (uint32_t)(ch_mode == FHIP_CH_NOT_STEREO ? nch - 1 : ch_mode));
        hs.put(3, (uint32_t)bps_code);
        hs.put(1, 0);
        if (number < 0x80) {
            hs.put(8, number);
        } else {
            const int bytes = (ilog2_dev(number) + 4) / 5;          // encode.c:696-716
            int sh = (bytes - 1) * 6;
            hs.put(8, ((256u - (256u >> bytes)) | (number >> sh)) & 0xFFu);
            while (sh >= 6) { sh -= 6; hs.put(8, 0x80u | ((number >> sh) & 0x3Fu)); }
        }
        if (bs1 >= 0) hs.put(bs1 < 256 ? 8 : 16, (uint32_t)bs1);
        if (sr_code1 > 0) hs.put(sr_code1 < 256 ? 8 : 16, (uint32_t)sr_code1);
        uint8_t c8 = 0;
        for (int q = 0; q < (hs.bits() >> 3); q++) {
            c8 ^= s_hdr[q];
            for (int b = 0; b < 8; b++) c8 = (uint8_t)((c8 & 0x80) ? ((c8 << 1) ^ 0x07) : (c8 << 1));
        }
        hs.put(8, c8);
    }
    // ---- per-subframe prefixes (encode.c:871-905, 800-869), one thread each --
    if (tid < nch) {
        const fhip_subframe_info *i = &fi[tid];
        const int type = verbatim ? FHIP_SUB_VERBATIM : s_info[tid][0];
        const int order = s_info[tid][2], obits = s_info[tid][4], wasted = s_info[tid][5];
        MiniSink ps{s_prefix[tid], 0};
        ps.put(1, 0);
        ps.put(6, (uint32_t)(verbatim ? FHIP_SUB_VERBATIM : s_info[tid][1]));
        if (wasted) { ps.put(1, 1); ps.put(wasted - 1, 0); ps.put(1, 1); }
        else ps.put(1, 0);
        const uint32_t omask = (obits >= 32) ? 0xFFFFFFFFu : ((1u << obits) - 1u);
        if (type == FHIP_SUB_CONSTANT) {
            ps.put(obits, (uint32_t)i->warmup[0] & omask);
        } else if (type == FHIP_SUB_FIXED || type == FHIP_SUB_LPC) {
            for (int t = 0; t < order; t++) ps.put(obits, (uint32_t)i->warmup[t] & omask);
            if (type == FHIP_SUB_LPC) {
                ps.put(4, (uint32_t)(P.lpc_precision - 1));
                ps.put(5, (uint32_t)s_info[tid][3] & 31u);
                const uint32_t cmask = (1u << P.lpc_precision) - 1u;
                for (int t = 0; t < order; t++) ps.put(P.lpc_precision, (uint32_t)i->coefs[t] & cmask);
            }
        }
        s_info[tid][0] = type;
        s_info[tid][3] = ps.bits();             // reuse: prefix length
    }
    __syncthreads();
    if (tid == 0) {
        int ns = 0;
        long long pos = 0;
        s_seg[ns++] = AsmSeg{0, -1, s_hdr_bits, pos}; pos += s_hdr_bits;
        for (int c = 0; c < nch; c++) {
            s_seg[ns++] = AsmSeg{0, c, s_info[c][3], pos}; pos += s_info[c][3];
            const int type = s_info[c][0];
            if (type == FHIP_SUB_VERBATIM) {
                const long long nb = (long long)n * s_info[c][4];
                s_seg[ns++] = AsmSeg{2, c, nb, pos}; pos += nb;
            } else if (type == FHIP_SUB_FIXED || type == FHIP_SUB_LPC) {
                s_seg[ns++] = AsmSeg{1, c, s_info[c][6], pos}; pos += s_info[c][6];
            }
        }
        s_nseg = ns;
        s_total_bits = pos;
    }
    __syncthreads();

    const long long total_bits = s_total_bits;
    const int body_bytes = (int)((total_bits + 7) >> 3);          // before the CRC-16
    const int nwords = (body_bytes + 3) >> 2;
    const int nseg = s_nseg;

    // ---- every output dword from the segments that overlap it ----------------
    for (int w = tid; w < nwords; w += NT) {
        const long long w0 = (long long)w * 32, w1 = w0 + 32;
        uint32_t word = 0;
        // the segments lie back to back in ascending order: skip those that end at or before this
        // word (one 8-byte LDS read each), stop at the first that starts behind it -- most words lie
        // inside one long residual section
        int q0 = 0;
        while (q0 < nseg - 1 && s_seg[q0 + 1].dst <= w0) q0++;
        for (int q = q0; q < nseg; q++) {
            const AsmSeg sg = s_seg[q];
            if (sg.dst >= w1) break;
            const long long a = max(sg.dst, w0), b = min(sg.dst + sg.nbits, w1);
            if (a >= b) continue;
            const int cnt = (int)(b - a);
            const long long sp = a - sg.dst;                      // bit offset inside the segment
            uint32_t bitsv;                                       // cnt bits, right aligned
            if (sg.kind == 2) {
                const int c = sg.ch, obits = s_info[c][4], wasted = s_info[c][5], cm = s_info[c][7];
                const uint32_t omask = (obits >= 32) ? 0xFFFFFFFFu : ((1u << obits) - 1u);
                int i = (int)(sp / obits), offb = (int)(sp % obits), got = 0;
                unsigned long long acc = 0;
                while (got < cnt) {
                    const uint32_t v = (uint32_t)asm_sample(pcm_frame, nch, c, i, cm, wasted) & omask;
                    const int take = min(obits - offb, cnt - got);
                    const uint32_t piece = (take >= 32) ? v : ((v >> (obits - offb - take)) & ((1u << take) - 1u));
                    acc = (acc << take) | piece;
                    got += take; offb = 0; i++;
                }
                bitsv = (uint32_t)acc;
            } else {
                // 64 source bits that start at the dword holding bit sp
                const long long sw = sp >> 5;
                uint32_t hi, lo;
                if (sg.kind == 0) {
                    const uint8_t *src = (sg.ch < 0) ? s_hdr : s_prefix[sg.ch];
                    const int lim = (sg.ch < 0) ? 32 : ASM_PREFIX_BYTES;
                    uint32_t bv[8];
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const long long bi = sw * 4 + z;
                        bv[z] = (bi < lim) ? src[bi] : 0u;
                    }
                    hi = (bv[0] << 24) | (bv[1] << 16) | (bv[2] << 8) | bv[3];
                    lo = (bv[4] << 24) | (bv[5] << 16) | (bv[6] << 8) | bv[7];
                } else {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(
                        rice + ((size_t)f * nch + sg.ch) * (size_t)slot_bytes);
                    hi = __builtin_bswap32(src[sw]);
                    lo = ((sw + 1) * 4 < slot_bytes) ? __builtin_bswap32(src[sw + 1]) : 0u;
                }
                const unsigned long long x = ((unsigned long long)hi << 32) | lo;
                const int sh = (int)(sp & 31);
                bitsv = (uint32_t)((x << sh) >> (64 - cnt));
            }
            word |= bitsv << (32 - (int)(a - w0) - cnt);
        }
        out32[w] = __builtin_bswap32(word);
    }
    __syncthreads();                       // the frame body is in memory (same CU)

    // ---- CRC-16 (crc.c:59-94): ONE wave, 64 chunks, merged by shuffles -----------
    // (round 3.  Round 2 gave every thread a chunk and merged 256 partial CRCs through LDS: a
    // square-and-multiply for x^(8L) and eight barrier-separated steps with two 16-step GF(2)
    // products each -- ~2600 of the kernel's ~6000 instructions per wave, on all four waves, for
    // a frame of a few KB.)  Lane i >= 1 takes the L bytes that end (63 - i) L bytes before the
    // body's end, L the smallest power of two with 64 L >= body; lane 0 takes what is in front
    // (lanes whose range lies before the body are empty: crc 0).  crc(A || B) = crc(A) x^(8 |B|) +
    // crc(B) only asks for the RIGHT part's length, so every merge constant is x^(8 L 2^j) mod P, a
    // compile-time table entry.
    if (tid >= WAVE) return;
    {
        int k = 2;
        while ((64 << k) < body_bytes) k++;
        const int L = 1 << k;
        const int e = body_bytes - (63 - tid) * L;             // end of this lane's chunk
        const int s0 = max(e - L, 0);
        uint16_t c = 0;
        uint32_t wv = 0;
        // (the bytes were stored by other lanes of this workgroup: read past L1)
        int bi = s0;
        for (; bi < e && (bi & 3) != 0; bi++) {                  // up to the first whole word
            if (bi == s0) wv = __hip_atomic_load(&out32[bi >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t byte = (wv >> (8 * (bi & 3))) & 0xFFu;
            c = (uint16_t)((c << 8) ^ s_crc_tab[((c >> 8) ^ byte) & 0xFFu]);
        }
        // four bytes a step: the state only meets the first two, the four table reads are independent
        // (a byte a step is a dependent LDS read per byte: 512 in a row for a 20 KB frame)
        for (; bi + 4 <= e; bi += 4) {
            wv = __hip_atomic_load(&out32[bi >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t x0 = ((c >> 8) ^ wv) & 0xFFu, x1 = (c ^ (wv >> 8)) & 0xFFu;
            c = (uint16_t)(s_crc_zk[2][x0] ^ s_crc_zk[1][x1] ^ s_crc_zk[0][(wv >> 16) & 0xFFu] ^ s_crc_tab[wv >> 24]);
        }
        for (; bi < e; bi++) {
            if ((bi & 3) == 0) wv = __hip_atomic_load(&out32[bi >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t byte = (wv >> (8 * (bi & 3))) & 0xFFu;
            c = (uint16_t)((c << 8) ^ s_crc_tab[((c >> 8) ^ byte) & 0xFFu]);
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            // lanes = 0 mod 2^(j+1) absorb the node 2^j lanes to their right (2^j chunks of L bytes)
            const uint16_t right = (uint16_t)__shfl_down((int)c, 1 << j, WAVE);
            c = (uint16_t)(crc16_mulmod(c, crc16_pow8(k + j)) ^ right);
        }
        if (tid == 0) {
            out[body_bytes] = (uint8_t)(c >> 8);
            out[body_bytes + 1] = (uint8_t)c;
            frame_bytes[f] = body_bytes + 2;
        }
    }
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
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(NT), 0, st, p, n, pcm, info, rice,
                       (long long)slot_bytes, frames, (long long)frame_stride, frame_bytes,
                       number_base, number_step, numbers, sr0, sr1, bpsc, vsize, frame_src, dev_frames, MultiBin{});
    return hipGetLastError();
}

hipError_t launch_assemble_bins(hipStream_t st, const fhip_params &p, const MultiBin &mb, const int32_t *pcm,
                                const fhip_subframe_info *info, const uint8_t *rice, uint8_t *frames,
                                int32_t *frame_bytes, const uint32_t *numbers, const long long *frame_src)
{
    if (mb.nbins < 1 || !numbers || !frame_src) return hipErrorInvalidValue;
    const int slots = mb.unit0[mb.nbins - 1] + mb.cap[mb.nbins - 1];
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
    hipLaunchKernelGGL(k_assemble, dim3(slots), dim3(NT), 0, st, p, 0, pcm, info, rice, 0ll, frames, 0ll,
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
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int eighth = block_size / 8;
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
            bin[q] = (q < f) ? len[q] / eighth - 1 : -1;
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
                    const int slot = bins.slot0[k] + j;
                    order[run[8]++] = slot;
                    frame_src[slot] = pos * nch;
                    src_off[slot] = bins.fr_off[k] + (long long)j * bins.stride[k];
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
    const int f = blockIdx.x, tid = threadIdx.x;
    if (f >= dev_count(dev_frames, 0)) return;
    const int slot = order[f];
    const int len = max(fbytes[slot], 0);
    if (tid == 0 && stream_bytes) stream_bytes[f] = fbytes[slot];   // the frame sizes in stream order
    if (offsets[f] + len > cap) return;                             // the caller's buffer ends here (totals[3])
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
