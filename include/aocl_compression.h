/* include/aocl_compression.h -- the reference's unified API, as exported by libaocl_compression_gpu.so.
 *
 * Declarations mirror /root/reference/api/aocl_compression.h:95-231 (AOCL-Compression 4.2.0): same enum
 * values, same LP64 struct layout, same five functions with the same argument meaning and error
 * behaviour (api/api.cpp:45-189), so a program written against the reference header can be relinked
 * against this library.  LZ4, SNAPPY and ZSTD (levels 1-4 and every negative level: the "fast" and "double-fast" strategies, level 0 = 3,
 * inputs of any size) have a GPU backend (LZ4HC: decompression only - its streams are LZ4 blocks, api/codec.cpp:179-188;
 * compression fails);
 * other enumerators report ERR_EXCLUDED_METHOD from aocl_llc_setup exactly as a reference build
 * configured with -DAOCL_EXCLUDE_<codec>=ON does (api/api.cpp:133-162).
 */
#ifndef AOCL_COMPRESSION_GPU_UNIFIED_H
#define AOCL_COMPRESSION_GPU_UNIFIED_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AOCL_COMPRESSION_LIBRARY_VERSION "AOCL-Compression 4.2.0"

typedef enum {
    ERR_INVALID_INPUT = -5,
    ERR_UNSUPPORTED_METHOD,
    ERR_EXCLUDED_METHOD,
    ERR_COMPRESSION_FAILED,
    ERR_COMPRESSION_INVALID_OUTPUT
} aocl_error_type;

typedef enum { LZ4 = 0, LZ4HC, LZMA, BZIP2, SNAPPY, ZLIB, ZSTD, AOCL_COMPRESSOR_ALGOS_NUM } aocl_compression_type;

typedef struct {
    char* inBuf;      /* input buffer (host) */
    char* outBuf;     /* output buffer (host) */
    char* workBuf;    /* codec context created by aocl_llc_setup */
    size_t inSize;
    size_t outSize;
    size_t level;
    size_t optVar;
    int numThreads;
    int numMPIranks;
    size_t memLimit;
    int measureStats; /* 1 => fill cSize/cTime/cSpeed or dSize/dTime/dSpeed (ns, bytes*1000/ns) */
    uint64_t cSize, dSize, cTime, dTime;
    float cSpeed, dSpeed;
    int optOff;
    int optLevel;
} aocl_compression_desc;

int64_t aocl_llc_compress(aocl_compression_desc* handle, aocl_compression_type codec_type);
int64_t aocl_llc_decompress(aocl_compression_desc* handle, aocl_compression_type codec_type);
int32_t aocl_llc_setup(aocl_compression_desc* handle, aocl_compression_type codec_type);
void aocl_llc_destroy(aocl_compression_desc* handle, aocl_compression_type codec_type);
const char* aocl_llc_version(void);

#ifdef __cplusplus
}
#endif
#endif
