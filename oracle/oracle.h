/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's block-codec hot path
 * (amd/aocl-compression 4.2.0).  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check the HIP kernels; the product library (libaocl_compression_gpu.so)
 * never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  Every function here is checked byte-for-byte against the real reference
 * compiled in place (oracle/_ref/libaocl_ref.so, see build_ref.sh) by tests/test_oracle_vs_ref.py,
 * and against the committed golden vectors in tests/golden/ (generated from that reference build by
 * tests/golden/make_golden*.py).  That includes the whole-input paths: Zstd frames of several blocks
 * (orc_zstd_compress_frame_cp against ZSTD_compress_advanced, fast and double-fast, any window) and the blocks of an
 * LZ4 frame with linked blocks (orc_lz4_compress_linked_blocks against LZ4F_compressFrame).
 */
#ifndef AOCL_GPU_ORACLE_H
#define AOCL_GPU_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- LZ4 block format, level 1 (reference: algos/lz4/lz4.c) ---- */
int orc_lz4_compress_bound(int srcSize);                               /* lz4.h:261 LZ4_COMPRESSBOUND */
/* == LZ4_compress_default (lz4.c:2967-2975): returns compressed size, 0 on failure. */
int orc_lz4_compress(const uint8_t* src, int srcSize, uint8_t* dst, int dstCap);
/* == LZ4_compress_fast (lz4.c:2937-2965) */
int orc_lz4_compress_fast(const uint8_t* src, int srcSize, uint8_t* dst, int dstCap, int acceleration);
/* the blocks of LZ4F_compressFrame with linked blocks (lz4frame.c:373-419, lz4.c:3179-3256 in prefix mode): see lz4_oracle.c */
int orc_lz4_compress_linked_blocks(const uint8_t* src, size_t n, size_t blockSize, int acceleration, uint8_t* dst, size_t dstCap,
                                   uint32_t* blockLen);
/* == LZ4_decompress_safe (lz4.c:4898-4923): returns decoded size, <0 on malformed input. */
int orc_lz4_decompress(const uint8_t* src, int srcSize, uint8_t* dst, int dstCap);

/* ---- Snappy raw format (reference: algos/snappy/snappy.cc) ---- */
size_t orc_snappy_max_compressed_length(size_t n);                     /* snappy.cc:160-182 */
/* == snappy::RawCompress (snappy.cc:2494): returns compressed size. dst must hold max_compressed_length. */
size_t orc_snappy_compress(const uint8_t* src, size_t n, uint8_t* dst);
/* == snappy::GetUncompressedLength: 1 on success */
int orc_snappy_uncompressed_length(const uint8_t* src, size_t n, size_t* result);
/* == snappy::RawUncompress: 1 on success, 0 on malformed input. dst must hold the header length. */
int orc_snappy_uncompress(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap);

/* ---- Zstd level 1, one-shot frame (reference: algos/zstd/lib) ---- */
size_t orc_zstd_compress_bound(size_t n);                              /* zstd.h:473 */
/* == ZSTD_compress(dst,cap,src,n,1) through the AOCL fast path for n <= 128 KiB blocks per frame.
 * returns compressed size, 0 on failure */
size_t orc_zstd_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap);
size_t orc_zstd_compress_level(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap, int level);   /* level 1 or 2 */
void orc_set_zstd_level(int level);                                      /* chunk_driver.c: the Zstd level of the chunk drivers */
/* == ZSTD_compress_advanced with these cParams on n <= 128 KiB: strategy 1 "fast" (zstd_fast.c:427-706) or 2 "double-fast"
 * (zstd_double_fast.c:332-573; hashLog = long table, chainLog = short table, minMatch = short-hash bytes) */
size_t orc_zstd_compress_cparams(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap, int strategy,
                                 uint32_t hashLog, uint32_t chainLog, uint32_t minMatch);
/* == ZSTD_compress_advanced with "fast" parameters on an input of any size below 2 GiB: one frame of 128 KiB blocks
 * (zstd_compress.c:4504-4583) */
size_t orc_zstd_compress_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap,
                               uint32_t windowLog, uint32_t hashLog, uint32_t minMatch);
/* targetLength 0 or 1 (both mean step 2, zstd_fast.c:440); 1 also leaves the literals uncompressed (level -1, and row 0 of the
 * level tables) */
size_t orc_zstd_compress_frame_tl(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap,
                                  uint32_t windowLog, uint32_t hashLog, uint32_t minMatch, uint32_t targetLength);
/* any size, strategy 1 (fast) or 2 (double-fast: hashLog = long table, chainLog = short table) */
size_t orc_zstd_compress_frame_cp(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap, int strategy, uint32_t windowLog,
                                  uint32_t hashLog, uint32_t chainLog, uint32_t minMatch, uint32_t targetLength);
/* == ZSTD_decompress: returns decoded size, (size_t)-1 on error */
size_t orc_zstd_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap);

/* ---- chunked drivers (OpenMP over independent chunks; mirrors the reference's RAP partition loop,
 *      lz4.c:2684-2731) used for the cpu_baseline "port" timing and multi-chunk parity ---- */
/* codec: 0 lz4, 1 snappy, 2 zstd.  Chunk i covers [i*chunk, min(n,(i+1)*chunk)); output slot i starts
 * at i*slot.  out_len[i] receives the compressed size (0 = failure).  Returns number of failures. */
int orc_compress_chunks(int codec, const uint8_t* src, size_t n, size_t chunk,
                        uint8_t* dst, size_t slot, uint32_t* out_len, int threads);
int orc_decompress_chunks(int codec, const uint8_t* src, size_t slot, const uint32_t* in_len, size_t nchunks,
                          uint8_t* dst, size_t chunk, size_t n, int threads);
#ifdef __cplusplus
}
#endif
#endif
