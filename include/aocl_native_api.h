/* include/aocl_native_api.h -- the per-codec native entry points of amd/aocl-compression 4.2.0 that sit on the hot
 * path, exported by libaocl_compression_gpu.so with the reference's names, argument meaning and return conventions
 * (the reference .so is also linked as liblz4 / libsnappy / libzstd, CMakeLists.txt:616-635).
 *
 * Buffers are HOST buffers exactly as in the reference; one call = one input on the GPU (upload, one wavefront per
 * chunk / frame, download).  Bulk work should use the device-resident batch API of aocl_gpu_codec.h.  There is no CPU
 * fallback: without a gfx950 device compression returns the codec's failure value.
 *
 * Scope is the north-star path and the rows next to it: LZ4 (any acceleration) and its frame format with linked or
 * independent blocks, Snappy, Zstd with the "fast" strategy (levels 1-2 and every negative level) and the
 * "double-fast" strategy (levels 3-4; level 0 = 3) at any input size; anything else fails loudly with the codec's
 * error convention instead of silently doing something different (INTEGRATION.md lists every deviation from the
 * reference).
 */
#ifndef AOCL_NATIVE_API_H
#define AOCL_NATIVE_API_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- LZ4 1.9.3 (algos/lz4/lz4.h:128,137,203,232,287,309,464,470) ---- */
int         LZ4_versionNumber(void);                       /* 10903 */
const char* LZ4_versionString(void);                       /* "1.9.3" */
int         LZ4_compressBound(int inputSize);              /* 0 if inputSize > 0x7E000000 */
int         LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity);   /* 0 = failure */
int         LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration); /* any acceleration (clamped to 1..65537) */
int         LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);    /* malformed: -(ip - src) - 1 at the reference's point of failure (lz4.c:3799-3802), the same value */
char*       aocl_setup_lz4(int optOff, int optLevel, size_t insize, size_t level, size_t windowLog); /* NULL */
void        aocl_destroy_lz4(void);

/* ---- LZ4 frame format (algos/lz4/lz4frame.h:100-260, 380-470; lz4frame.c:373-460, 1398-1860) on the GPU block codec.
 *      Types mirror lz4frame.h.  LZ4F_compressFrame: the fast compressor only (compressionLevel < 3; negative = acceleration),
 *      the reference's bytes for LZ4F_blockLinked (the format's default: the blocks of a frame are one wavefront's work, in
 *      order, with the 64 KiB history across blocks) and for LZ4F_blockIndependent alike; aoclgpu_lz4f_set_block_linking(0) /
 *      AOCLGPU_LZ4F_LINKED=independent writes independent blocks whatever the preferences ask (throughput instead of the
 *      reference's bytes).  The streaming calls (LZ4F_compressBegin / Update / End) write independent blocks.
 *      LZ4F_decompress: independent and linked blocks, all checksums verified; whole frames are decoded in place, frames fed in
 *      pieces are collected and decoded once complete. ---- */
typedef size_t LZ4F_errorCode_t;
typedef enum { LZ4F_default = 0, LZ4F_max64KB = 4, LZ4F_max256KB = 5, LZ4F_max1MB = 6, LZ4F_max4MB = 7 } LZ4F_blockSizeID_t;
typedef enum { LZ4F_blockLinked = 0, LZ4F_blockIndependent } LZ4F_blockMode_t;
typedef enum { LZ4F_noContentChecksum = 0, LZ4F_contentChecksumEnabled } LZ4F_contentChecksum_t;
typedef enum { LZ4F_noBlockChecksum = 0, LZ4F_blockChecksumEnabled } LZ4F_blockChecksum_t;
typedef enum { LZ4F_frame = 0, LZ4F_skippableFrame } LZ4F_frameType_t;
typedef struct {
    LZ4F_blockSizeID_t blockSizeID; LZ4F_blockMode_t blockMode; LZ4F_contentChecksum_t contentChecksumFlag; LZ4F_frameType_t frameType;
    unsigned long long contentSize; unsigned dictID; LZ4F_blockChecksum_t blockChecksumFlag;
} LZ4F_frameInfo_t;
typedef struct { LZ4F_frameInfo_t frameInfo; int compressionLevel; unsigned autoFlush; unsigned favorDecSpeed; unsigned reserved[3]; } LZ4F_preferences_t;
typedef struct { unsigned stableDst; unsigned reserved[3]; } LZ4F_decompressOptions_t;
typedef struct { unsigned stableSrc; unsigned reserved[3]; } LZ4F_compressOptions_t;
typedef struct LZ4F_dctx_s LZ4F_dctx;
typedef struct LZ4F_cctx_s LZ4F_cctx;
unsigned    LZ4F_isError(LZ4F_errorCode_t code);
const char* LZ4F_getErrorName(LZ4F_errorCode_t code);
unsigned    LZ4F_getVersion(void);
int         LZ4F_compressionLevel_max(void);
size_t      LZ4F_compressFrameBound(size_t srcSize, const LZ4F_preferences_t* preferencesPtr);
size_t      LZ4F_compressFrame(void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize, const LZ4F_preferences_t* preferencesPtr);
/* streaming compression (lz4frame.c:570-1030): the reference's buffering rules, one launch per call for the blocks it completes */
LZ4F_errorCode_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version);
LZ4F_errorCode_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx);
size_t      LZ4F_compressBound(size_t srcSize, const LZ4F_preferences_t* preferencesPtr);
size_t      LZ4F_compressBegin(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* preferencesPtr);
size_t      LZ4F_compressUpdate(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                                const LZ4F_compressOptions_t* cOptPtr);
size_t      LZ4F_flush(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr);
size_t      LZ4F_compressEnd(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr);
LZ4F_errorCode_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version);
LZ4F_errorCode_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx);
void        LZ4F_resetDecompressionContext(LZ4F_dctx* dctx);
size_t      LZ4F_headerSize(const void* src, size_t srcSize);
size_t      LZ4F_getFrameInfo(LZ4F_dctx* dctx, LZ4F_frameInfo_t* frameInfoPtr, const void* srcBuffer, size_t* srcSizePtr);
size_t      LZ4F_decompress(LZ4F_dctx* dctx, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                            const LZ4F_decompressOptions_t* dOptPtr);

/* ---- Snappy C binding (algos/snappy/snappy-c.h:28-138) ---- */
typedef enum { SNAPPY_OK = 0, SNAPPY_INVALID_INPUT = 1, SNAPPY_BUFFER_TOO_SMALL = 2 } snappy_status;
snappy_status snappy_compress(const char* input, size_t input_length, char* compressed, size_t* compressed_length);
snappy_status snappy_uncompress(const char* compressed, size_t compressed_length, char* uncompressed, size_t* uncompressed_length);
size_t        snappy_max_compressed_length(size_t source_length);
char*         aocl_setup_snappy(int optOff, int optLevel, size_t insize, size_t level, size_t windowLog);   /* NULL (snappy.h:476) */
void          aocl_destroy_snappy(void);
snappy_status snappy_uncompressed_length(const char* compressed, size_t compressed_length, size_t* result);
snappy_status snappy_validate_compressed_buffer(const char* compressed, size_t compressed_length);

/* ---- Zstd 1.5.5 (algos/zstd/lib/zstd.h:184,201,261,291,353,423-452,487-489,542,581,606-615,657-764,2416-2435,3115-3166).
 *      GPU backend: the "fast" strategy (levels 1-2, every negative level, and through ZSTD_compress_advanced any "fast"
 *      parameter set) and the "double-fast" strategy (levels 3-4; level 0 = 3), with or without content size / content checksum.
 *      An input of ANY size is ONE frame, the reference's bytes: srcSize <= 128 KiB is a single block; a larger input is the
 *      reference's frame of several blocks with its window across blocks (zstd_compress.c:4504-4583), which is one wavefront's
 *      sequential work.  aoclgpu_zstd_set_large_input_mode / AOCLGPU_ZSTD_LARGE_INPUTS=pieces writes concatenated independent
 *      128 KiB frames instead (valid Zstandard at GiB/s, not the reference's bytes).  Levels 5-22, dictionaries and every other
 *      request return ZSTD_error_parameter_unsupported (see INTEGRATION.md for the full deviation list). ---- */
typedef enum { ZSTD_fast = 1, ZSTD_dfast = 2, ZSTD_greedy = 3, ZSTD_lazy = 4, ZSTD_lazy2 = 5, ZSTD_btlazy2 = 6, ZSTD_btopt = 7,
               ZSTD_btultra = 8, ZSTD_btultra2 = 9 } ZSTD_strategy;
typedef struct { unsigned windowLog, chainLog, hashLog, searchLog, minMatch, targetLength; ZSTD_strategy strategy; } ZSTD_compressionParameters;
typedef struct { int contentSizeFlag, checksumFlag, noDictIDFlag; } ZSTD_frameParameters;
typedef struct { ZSTD_compressionParameters cParams; ZSTD_frameParameters fParams; } ZSTD_parameters;
typedef enum { ZSTD_c_compressionLevel = 100, ZSTD_c_windowLog = 101, ZSTD_c_hashLog = 102, ZSTD_c_chainLog = 103, ZSTD_c_searchLog = 104,
               ZSTD_c_minMatch = 105, ZSTD_c_targetLength = 106, ZSTD_c_strategy = 107, ZSTD_c_contentSizeFlag = 200,
               ZSTD_c_checksumFlag = 201, ZSTD_c_dictIDFlag = 202 } ZSTD_cParameter;
typedef struct ZSTD_CCtx_s ZSTD_CCtx;
typedef struct ZSTD_DCtx_s ZSTD_DCtx;
unsigned    ZSTD_versionNumber(void);                      /* 10505 */
const char* ZSTD_versionString(void);                      /* "1.5.5" */
int         ZSTD_minCLevel(void);
int         ZSTD_maxCLevel(void);
int         ZSTD_defaultCLevel(void);
size_t      ZSTD_compressBound(size_t srcSize);
unsigned    ZSTD_isError(size_t code);
const char* ZSTD_getErrorName(size_t code);
size_t      ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
size_t      ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);   /* errors: the reference's code for the check that fails first (deviations: "Error VALUES") */
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize);
unsigned long long ZSTD_getDecompressedSize(const void* src, size_t srcSize);   /* 0 = unknown or error */
unsigned long long ZSTD_decompressBound(const void* src, size_t srcSize);       /* over all frames of the stream */
size_t      ZSTD_findFrameCompressedSize(const void* src, size_t srcSize);
typedef enum { ZSTD_reset_session_only = 1, ZSTD_reset_parameters = 2, ZSTD_reset_session_and_parameters = 3 } ZSTD_ResetDirective;
size_t      ZSTD_CCtx_reset(ZSTD_CCtx* cctx, ZSTD_ResetDirective reset);
size_t      ZSTD_DCtx_reset(ZSTD_DCtx* dctx, ZSTD_ResetDirective reset);
size_t      ZSTD_compress2(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);   /* level / frame flags from the context */
ZSTD_CCtx*  ZSTD_createCCtx(void);
size_t      ZSTD_freeCCtx(ZSTD_CCtx* cctx);
size_t      ZSTD_compressCCtx(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
ZSTD_DCtx*  ZSTD_createDCtx(void);
size_t      ZSTD_freeDCtx(ZSTD_DCtx* dctx);
size_t      ZSTD_decompressDCtx(ZSTD_DCtx* dctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);
size_t      ZSTD_CCtx_setParameter(ZSTD_CCtx* cctx, ZSTD_cParameter param, int value);   /* level and frame flags; others unsupported */
ZSTD_compressionParameters ZSTD_getCParams(int compressionLevel, unsigned long long estimatedSrcSize, size_t dictSize);
ZSTD_parameters ZSTD_getParams(int compressionLevel, unsigned long long estimatedSrcSize, size_t dictSize);  /* all 22 levels, both tables */
size_t      ZSTD_compress_advanced(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                   const void* dict, size_t dictSize, ZSTD_parameters params);
char*       aocl_setup_zstd_encode(int optOff, int optLevel, size_t insize, size_t level, size_t windowLog); /* NULL; optOff picks the upstream level table */
char*       aocl_setup_zstd_decode(int optOff, int optLevel, size_t insize, size_t level, size_t windowLog); /* NULL */
void        aocl_destroy_zstd_encode(void);
void        aocl_destroy_zstd_decode(void);

/* ---- Block-level external sequence producer for libzstd (algos/zstd/lib/zstd.h:4065-4098): has the ZSTD_sequenceProducer_F
 *      signature (outSeqs is a ZSTD_Sequence*); a user of upstream / reference libzstd registers it with
 *      ZSTD_registerSequenceProducer(cctx, NULL, (ZSTD_sequenceProducer_F*)aoclgpu_zstd_sequence_producer)
 *      and the GPU match finder (the level-1 "fast" parse whatever compressionLevel says, one block at a time, no history
 *      across blocks) feeds zstd's own entropy stage.
 *      Offsets are real offsets: set ZSTD_c_searchForExternalRepcodes = ZSTD_ps_enable so that zstd turns them back into repcodes
 *      (it only does so by default from level 10 on).
 *      Returns the number of sequences incl. the closing block delimiter, or ZSTD_SEQUENCE_PRODUCER_ERROR ((size_t)-1). ---- */
size_t aoclgpu_zstd_sequence_producer(void* state, void* outSeqs, size_t outSeqsCapacity, const void* src, size_t srcSize,
                                      const void* dict, size_t dictSize, int compressionLevel, size_t windowSize);

/* ---- RAP frame helpers (api/aocl_threads.h:109,133; threads/threads.c:315-336) ---- */
int32_t aocl_get_rap_frame_bound_mt(void);                 /* 16 + 12 * max partitions (65535 for the GPU build) */
int32_t aocl_skip_rap_frame_mt(char* src, int32_t src_size); /* 0: no RAP frame; > 0: its length; ERR_INVALID_INPUT on NULL */

#ifdef __cplusplus
}
#include <string>

/* ---- Snappy C++ binding (algos/snappy/snappy.h:255,280,369,391; snappy.cc:2271,2494) ---- */
namespace snappy {
void   RawCompress(const char* input, size_t input_length, char* compressed, size_t* compressed_length);
bool   RawUncompress(const char* compressed, size_t compressed_length, char* uncompressed);
size_t MaxCompressedLength(size_t source_bytes);
bool   GetUncompressedLength(const char* compressed, size_t compressed_length, size_t* result);
bool   IsValidCompressedBuffer(const char* compressed, size_t compressed_length);
size_t Compress(const char* input, size_t input_length, std::string* compressed);        /* snappy.h:268 */
bool   Uncompress(const char* compressed, size_t compressed_length, std::string* uncompressed);   /* snappy.h:282 */
}
#endif
#endif
