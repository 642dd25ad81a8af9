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

/* Deterministic synthetic PCM, channel-interleaved int32 as
 * flake_encode_frame() expects: nframes blocks of n samples per channel,
 * starting at absolute frame index first_frame. */
FLAKE_AMD_API void flake_amd_synth_pcm(int32_t *pcm, int64_t first_frame, int nframes,
                                       int n, int channels, int bps);

#ifdef __cplusplus
}
#endif
#endif /* FLAKE_AMD_H */
