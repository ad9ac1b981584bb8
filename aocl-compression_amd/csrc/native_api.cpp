// csrc/native_api.cpp -- the reference's per-codec native entry points on top of the GPU engine
// (include/aocl_native_api.h).  Host buffers, one chunk per call, no CPU fallback.
#include "../../include/aocl_native_api.h"
#include "../../include/aocl_gpu_codec.h"
#include <stdlib.h>
#include <string.h>
#include <string>

namespace {
const int kErrInvalidInput = -5;                          // api/aocl_compression.h:97 ERR_INVALID_INPUT

// varint32 length prefix of a Snappy stream (snappy.cc:1202-1223, snappy-stubs-internal.h:440)
bool sn_parse_len(const char* p, size_t n, uint32_t* out)
{
    uint32_t v = 0;
    for (uint32_t shift = 0, i = 0; shift <= 28; shift += 7, ++i) {
        if (i >= n) return false;
        const uint32_t c = (uint8_t)p[i];
        if (shift == 28 && (c & 0x7F) > 15) return false;  // LeftShiftOverflows
        v |= (c & 0x7F) << shift;
        if (c < 128) { *out = v; return true; }
    }
    return false;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- LZ4 ----------------------------------------
int LZ4_versionNumber(void) { return 10903; }
const char* LZ4_versionString(void) { return "1.9.3"; }
int LZ4_compressBound(int isize) { return (unsigned)isize > 0x7E000000u ? 0 : isize + isize / 255 + 16; }   // lz4.h:261

int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration)
{   // lz4.c:2937-2975 (AOCL adds the NULL checks); acceleration < 1 -> 1, > 65537 -> 65537 (lz4.c:2521-2522)
    if ((!src && srcSize != 0) || !dst || srcSize < 0 || dstCapacity < 0) return 0;
    if ((unsigned)srcSize > 0x7E000000u) return 0;
    const int64_t r = aoclgpu_compress_host_ex(AOCLGPU_LZ4, src, (size_t)srcSize, dst, (size_t)dstCapacity, AOCLGPU_LZ4_ACCELERATION(acceleration));
    return r > 0 && r <= 0x7FFFFFFF ? (int)r : 0;
}
int LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity)
{
    return LZ4_compress_fast(src, dst, srcSize, dstCapacity, 1);
}
int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity)
{   // lz4.c:4898-4923
    if (!src || !dst || compressedSize < 0 || dstCapacity < 0) return -1;
    const int64_t r = aoclgpu_decompress_host(AOCLGPU_LZ4, src, (size_t)compressedSize, dst, (size_t)dstCapacity);
    if (r >= 0) return (int)r;
    return r > -0x7FFFFFFF ? (int)r : -1;
}
char* aocl_setup_lz4(int, int, size_t, size_t, size_t) { return nullptr; }   // lz4.h:464-470: no host ISA variant to pick
void aocl_destroy_lz4(void) {}
char* aocl_setup_snappy(int, int, size_t, size_t, size_t) { return nullptr; }   // snappy.h:476-482, snappy.cc:3153-3184
void aocl_destroy_snappy(void) {}

// ---------------------------------------------------------------- Snappy (C) ---------------------------------
size_t snappy_max_compressed_length(size_t n) { return 32 + n + n / 6; }                 // snappy.cc:160-182
snappy_status snappy_compress(const char* input, size_t input_length, char* compressed, size_t* compressed_length)
{   // snappy-c.cc:37-49
    if (!compressed_length || (!input && input_length) || !compressed) return SNAPPY_INVALID_INPUT;
    if (*compressed_length < snappy_max_compressed_length(input_length)) return SNAPPY_BUFFER_TOO_SMALL;
    const int64_t r = aoclgpu_compress_host(AOCLGPU_SNAPPY, input, input_length, compressed, *compressed_length);
    if (r <= 0) return SNAPPY_INVALID_INPUT;
    *compressed_length = (size_t)r;
    return SNAPPY_OK;
}
snappy_status snappy_uncompressed_length(const char* compressed, size_t compressed_length, size_t* result)
{
    uint32_t v;
    if (!compressed || !result || !sn_parse_len(compressed, compressed_length, &v)) return SNAPPY_INVALID_INPUT;
    *result = v;
    return SNAPPY_OK;
}
snappy_status snappy_uncompress(const char* compressed, size_t compressed_length, char* uncompressed, size_t* uncompressed_length)
{   // snappy-c.cc:51-69
    size_t real;
    if (!uncompressed || !uncompressed_length) return SNAPPY_INVALID_INPUT;
    if (snappy_uncompressed_length(compressed, compressed_length, &real) != SNAPPY_OK) return SNAPPY_INVALID_INPUT;
    if (*uncompressed_length < real) return SNAPPY_BUFFER_TOO_SMALL;
    const int64_t r = aoclgpu_decompress_host(AOCLGPU_SNAPPY, compressed, compressed_length, uncompressed, real);
    if (r < 0 || (size_t)r != real) return SNAPPY_INVALID_INPUT;
    *uncompressed_length = real;
    return SNAPPY_OK;
}
snappy_status snappy_validate_compressed_buffer(const char* compressed, size_t compressed_length)
{
    size_t real;
    if (snappy_uncompressed_length(compressed, compressed_length, &real) != SNAPPY_OK) return SNAPPY_INVALID_INPUT;
    char* tmp = (char*)malloc(real ? real : 1);
    if (!tmp) return SNAPPY_INVALID_INPUT;
    const int64_t r = aoclgpu_decompress_host(AOCLGPU_SNAPPY, compressed, compressed_length, tmp, real);
    free(tmp);
    return r >= 0 && (size_t)r == real ? SNAPPY_OK : SNAPPY_INVALID_INPUT;
}

// ---------------------------------------------------------------- RAP frame ----------------------------------
int32_t aocl_get_rap_frame_bound_mt(void)
{   // threads/threads.c:315-318 with the thread count replaced by the GPU build's partition limit (the RAP
    // header stores the partition count in 16 bits, threads/threads.h:82-83)
    return 8 + 4 + 2 + 2 + 12 * 65535;
}
int32_t aocl_skip_rap_frame_mt(char* src, int32_t src_size)
{   // threads/threads.c:320-336
    if (!src) return kErrInvalidInput;
    int64_t magic = 0;
    if (src_size >= 8) memcpy(&magic, src, 8);
    if (src_size < 8 || magic != 0x434C4C5F4C434F41ll) return 0;
    if (src_size < 12) return 0;
    uint32_t len; memcpy(&len, src + 8, 4);
    return (int32_t)len;
}

}  // extern "C"

// ---------------------------------------------------------------- Snappy (C++) -------------------------------
namespace snappy {
size_t MaxCompressedLength(size_t n) { return snappy_max_compressed_length(n); }
void RawCompress(const char* input, size_t input_length, char* compressed, size_t* compressed_length)
{   // snappy.cc:2494-2503: NULL arguments -> silent return, *compressed_length untouched
    if (!compressed || !compressed_length || (!input && input_length)) return;
    const int64_t r = aoclgpu_compress_host(AOCLGPU_SNAPPY, input, input_length, compressed, MaxCompressedLength(input_length));
    if (r > 0) *compressed_length = (size_t)r;
}
bool GetUncompressedLength(const char* compressed, size_t compressed_length, size_t* result)
{
    return snappy_uncompressed_length(compressed, compressed_length, result) == SNAPPY_OK;
}
bool RawUncompress(const char* compressed, size_t compressed_length, char* uncompressed)
{   // snappy.cc:2271: the caller sized `uncompressed` from GetUncompressedLength
    size_t real;
    if (!uncompressed || !GetUncompressedLength(compressed, compressed_length, &real)) return false;
    const int64_t r = aoclgpu_decompress_host(AOCLGPU_SNAPPY, compressed, compressed_length, uncompressed, real);
    return r >= 0 && (size_t)r == real;
}
bool IsValidCompressedBuffer(const char* compressed, size_t compressed_length)
{
    return snappy_validate_compressed_buffer(compressed, compressed_length) == SNAPPY_OK;
}
size_t Compress(const char* input, size_t input_length, std::string* compressed)
{   // snappy.cc:2656-2680: the string is resized to the bound, filled, trimmed; returns the compressed size
    if (!compressed || (!input && input_length)) return 0;
    try { compressed->resize(MaxCompressedLength(input_length)); } catch (const std::exception&) { return 0; }
    size_t n = 0;
    RawCompress(input, input_length, &(*compressed)[0], &n);
    compressed->resize(n);
    return n;
}
bool Uncompress(const char* compressed, size_t compressed_length, std::string* uncompressed)
{   // snappy.cc:2392-2410
    size_t ulen;
    if (!uncompressed || !GetUncompressedLength(compressed, compressed_length, &ulen)) return false;
    try { uncompressed->resize(ulen); } catch (const std::exception&) { return false; }
    if (ulen == 0) { char tmp; return RawUncompress(compressed, compressed_length, &tmp); }
    return RawUncompress(compressed, compressed_length, &(*uncompressed)[0]);
}
}  // namespace snappy
