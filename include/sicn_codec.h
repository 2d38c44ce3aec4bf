/*
 * sicn_codec.h — latent container ("SICL" v1) and static rANS entropy coder on the GPU.
 *
 * EXTENSION BEYOND THE REFERENCE.  The reference never encodes its latent: `conv_3_out` is an
 * in-memory stream handed straight to layer 4 (conv_nonsquare_top.cpp:322-332) and the tree
 * contains no coder, CDF or bitstream of any kind.  These entry points therefore
 * replace NO reference interface; they implement SURVEY.md §8(f) rows 1-2 to this project's own
 * specification (oracle/sicn_codec_oracle.c states it in full).  Parity status: UNPINNED — the
 * tests show decode(encode(x)) == x and GPU output == CPU oracle output, byte for byte.
 *
 * Container: 48-byte header (magic "SICL", version, mode, image and latent dimensions, stream
 * count, payload size, adler32 of the latent), then for modes 2 and 3 a 128 x u16 frequency table and a
 * u32 byte count per stream, then the payload.  Symbols are latent bytes, which are < 128
 * (conv_nonsquare_top.cpp:273-275 zeroes every value with the MSB set).
 *   mode 0  raw8     payload = the latent
 *   mode 1  packed7  8 symbols -> 7 bytes
 *   mode 2  rANS     byte-wise rANS, 12-bit static frequencies measured on this latent, independent
 *                    streams of 1024 symbols (one GPU lane per stream), stream offsets by a
 *                    wavefront-level prefix scan
 *   mode 3  rANS-W   the wavefront form (the one to use): streams of 16384 symbols (or the encoder's choice of a shorter
 *                    power of two down to 1024: the _sl entry points), each coded by one wave
 *                    whose 64 lanes hold 64 interleaved rANS states sharing one stream of 16-bit words; a
 *                    lane's word position inside a step is a wavefront-level scan (popcount of a ballot).
 *                    Lane l owns 4 consecutive symbols of every 256-symbol block.  Same frequency table as
 *                    mode 2; on a 4K latent the kernels are 6-7x faster (≈55 us each way), +0.8 % bytes
 *
 * All pointers are DEVICE pointers unless named *_host.  Unlike sicn.h's launch functions, the plain
 * calls synchronise `hip_stream` (sizes have to come back to the caller); the *_async pair does not.
 */
#ifndef SICN_CODEC_H
#define SICN_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SICN_CODEC_RAW8 0
#define SICN_CODEC_PACKED7 1
#define SICN_CODEC_RANS 2
#define SICN_CODEC_RANSW 3
#define SICN_CODEC_RANSWC 4 /* rANS-W with the conditional (hyperprior + checkerboard context) model, see below */
#define SICN_CODEC_HEADER_BYTES 48
#define SICN_CODEC_STREAM_SYMBOLS 1024
#define SICN_CODEC_WSTREAM_SYMBOLS 16384 /* mode 3: 64 lanes x 256 steps */
#define SICN_EBADMSG (-74) /* checksum of the decoded latent does not match the header */

/* 44 bytes since library version 0.2 (sicn_version() >= 2): `stream_symbols` was appended in round 3 without a version bump
 *  — a caller built against the 40-byte struct must check sicn_version() before passing it to sicn_codec_parse_header
 * / sicn_codec_decode / sicn_codec_decode_batch, which write all 11 fields. */
typedef struct sicn_codec_info {
    uint32_t mode, image_width, image_height, lat_w, lat_h, lat_c, n_symbols, n_streams, payload_bytes, adler32;
    uint32_t stream_symbols; /* header dword 9: symbols per stream (mode 3: the encoder's choice, see the _sl entry points) */
} sicn_codec_info;

/* Upper bound of the container size / device scratch needed for n_symbols latent bytes.  sicn_codec_workspace_bytes(mode 3) covers
 * a container of ANY admissible stream length (the decoders read the length from the header), sicn_codec_max_bytes(mode 3) the
 * default length of 16384 (the _sl functions size a given length exactly). */
size_t sicn_codec_max_bytes(int mode, uint32_t n_symbols);
size_t sicn_codec_workspace_bytes(int mode, uint32_t n_symbols);

/* latent: [lat_h][lat_w][lat_c] uint8 (values < 128, else SICN_EINVAL).  Writes the container to
 * `out` (capacity >= sicn_codec_max_bytes) and its size to *out_bytes. */
int sicn_codec_encode(int mode, const uint8_t *latent, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                      uint32_t image_width, uint32_t image_height, uint8_t *out, size_t out_capacity,
                      size_t *out_bytes, void *workspace, size_t workspace_bytes, void *hip_stream);

/* Parses a container header from HOST memory (first SICN_CODEC_HEADER_BYTES bytes). */
int sicn_codec_parse_header(const uint8_t *header_host, size_t bytes, sicn_codec_info *info);

/* Decodes a container held in device memory into `latent` (capacity >= n_symbols) and verifies the
 * checksum (SICN_EBADMSG on mismatch, SICN_EINVAL on a malformed container). */
int sicn_codec_decode(const uint8_t *container, size_t bytes, uint8_t *latent, size_t latent_capacity,
                      sicn_codec_info *info_or_null, void *workspace, size_t workspace_bytes, void *hip_stream);

/* Batches (mode SICN_CODEC_RANSW only): n_images latents of one shape, `latents` = [n][lat_h][lat_w][lat_c], container i
 * at `out + i * slot_bytes` (slot_bytes >= sicn_codec_max_bytes, even), its size in out_bytes_host[i].  Byte-identical
 * to n single calls.  Wrappers around the asynchronous pair below: one host synchronisation per encode batch (the
 * sizes come back), two per decode batch (the shape is read from the headers first). */
size_t sicn_codec_batch_workspace_bytes(int mode, uint32_t n_symbols, uint32_t n_images);
int sicn_codec_encode_batch(int mode, const uint8_t *latents, uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                            uint32_t image_width, uint32_t image_height, uint8_t *out, size_t slot_bytes,
                            size_t *out_bytes_host, void *workspace, size_t workspace_bytes, void *hip_stream);
/* containers[i] at `containers + i * slot_bytes` with bytes_host[i] valid bytes -> latents + i * latent_stride. */
int sicn_codec_decode_batch(const uint8_t *containers, size_t slot_bytes, const size_t *bytes_host, uint32_t n_images,
                            uint8_t *latents, size_t latent_stride, sicn_codec_info *infos_or_null, void *workspace,
                            size_t workspace_bytes, void *hip_stream);

/* Asynchronous batch pair (rANS-W): everything — statistics, the 12-bit frequency table, header, streams, stream
 * offsets, compaction; header / table validation, streams, checksum — runs on the device.  The calls only enqueue
 * work on hip_stream: no host synchronisation, no allocation, capturable in a hipGraph.  Results are reported in
 * DEVICE memory: status[i].error == 0 on success (encode: bit 0 = a symbol >= 128, bit 1 = no valid frequency
 * table; decode: bits 2-6 = malformed container (SICN_EINVAL), bit 7 = checksum mismatch (SICN_EBADMSG), bit 8 =
 * slot shorter than its fixed part), status[i].bytes = container size (encode) / symbols decoded (decode).
 * Containers are byte-identical to sicn_codec_encode / sicn_codec_encode_batch and to the oracle.
 * The decoder takes the latent shape from the CALLER (a container whose header disagrees is an error) and, optionally,
 * the number of valid bytes of each slot from a device array (e.g. the encoder's status array); NULL = slot_bytes. */
typedef struct sicn_codec_status {
    uint32_t error;
    uint32_t bytes;
} sicn_codec_status;
int sicn_codec_encode_batch_async(const uint8_t *latents, uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                  uint32_t image_width, uint32_t image_height, uint8_t *out, size_t slot_bytes,
                                  sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes, void *hip_stream);
int sicn_codec_decode_batch_async(const uint8_t *containers, size_t slot_bytes, const sicn_codec_status *valid_dev_or_null,
                                  uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, uint8_t *latents,
                                  size_t latent_stride, sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes,
                                  void *hip_stream);

/* The same pair with the STREAM LENGTH as an encoder parameter (mode 3 containers carry it in header dword 9, so every decoder
 * of this library — and the oracle — reads containers of any admissible length): stream_symbols = a power of two, 1024 ..
 * 16384 (SICN_CODEC_WSTREAM_SYMBOLS, what the entry points above use).  A stream is a serial chain of stream_symbols / 64
 * steps on one wave: a 1080p latent (1.57 M symbols) is 96 streams of 16384 — 96 waves on 256 CUs — or 192 of 8192 at half
 * the chain, 765 of 2048 at an eighth; every stream ends with 260 bytes of flush (its 64 final states and its length entry), which
 * is inherent to 64 interleaved states (>= 16 bits each): + 2.8 % bytes at 8192 on that latent, + 8 % at 4096, + 14 % at 2048
 * (measured r03: 3.45 -> 3.94 bit / pixel).  The Python wrapper's default (codec.auto_stream_symbols) is 16384 for latents of at
 * least 128 such streams and 8192 below — a function of ONE image's latent, never of the batch.  Large batches gain nothing.
 * Size the slots and the workspace with the _sl functions; the decoder must be given the same length (a container whose
 * header disagrees is an error, bit 2). */
size_t sicn_codec_max_bytes_sl(uint32_t n_symbols, uint32_t stream_symbols);
size_t sicn_codec_workspace_bytes_sl(uint32_t n_symbols, uint32_t stream_symbols);
size_t sicn_codec_batch_workspace_bytes_sl(uint32_t n_symbols, uint32_t n_images, uint32_t stream_symbols);
int sicn_codec_encode_batch_async_sl(const uint8_t *latents, uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                     uint32_t image_width, uint32_t image_height, uint8_t *out, size_t slot_bytes,
                                     sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes, void *hip_stream,
                                     uint32_t stream_symbols);
int sicn_codec_decode_batch_async_sl(const uint8_t *containers, size_t slot_bytes, const sicn_codec_status *valid_dev_or_null,
                                     uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, uint8_t *latents,
                                     size_t latent_stride, sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes,
                                     void *hip_stream, uint32_t stream_symbols);

/* Mode 4, "rANS-WC": the hyperprior / context-model coder (BASELINE.json configs[4]; no reference
 * counterpart; specification: oracle/sicn_hyper_oracle.c).  Besides the latent both sides hold a SCALE MAP of the same
 * shape (values < 128; the hyper-synthesis output).  Every symbol is coded with one of 16 static tables carried in the
 * container; its class is `scale >> 3` for the anchor half of a checkerboard ((x + y) even) and
 * `min(15, ((scale >> 3) + (max of the 4 anchor neighbours >> 3) + 1) >> 1)` for the other half, which is therefore decoded
 * in a second pass.  Both passes are rANS-W streams (64 interleaved states per wave).  Asynchronous only: same status
 * conventions as the pair above; lat_c must be a multiple of 4, latents / scales 4-byte aligned, n images back to back
 * ([n][lat_h][lat_w][lat_c]). */
size_t sicn_codec_ctx_max_bytes(uint32_t lat_w, uint32_t lat_h, uint32_t lat_c);
size_t sicn_codec_ctx_workspace_bytes(uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, uint32_t n_images);
int sicn_codec_ctx_encode_batch_async(const uint8_t *latents, const uint8_t *scales, uint32_t n_images, uint32_t lat_w,
                                      uint32_t lat_h, uint32_t lat_c, uint32_t image_width, uint32_t image_height, uint8_t *out,
                                      size_t slot_bytes, sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes,
                                      void *hip_stream);
int sicn_codec_ctx_decode_batch_async(const uint8_t *containers, size_t slot_bytes, const sicn_codec_status *valid_dev_or_null,
                                      const uint8_t *scales, uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                      uint8_t *latents, sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes,
                                      void *hip_stream);

/* Self-test (host arithmetic only, no GPU): checks the rANS-W encoder's reciprocal divide against x / f for
 * f in [f_begin, f_end) over the states the encoder can hold. Returns the number of wrong results (0 = pass). */
long long sicn_codec_selftest_div(uint32_t f_begin, uint32_t f_end, unsigned long long *n_checked);

#ifdef __cplusplus
}
#endif
#endif /* SICN_CODEC_H */
