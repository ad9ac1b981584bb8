/* oracle/chunk_driver.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * OpenMP drivers that run a one-shot codec function over independent chunks -- the decomposition of the
 * reference's RAP partition loops (lz4.c:2684-2731, snappy.cc:2533-2562, zstd_compress.c:5443-5489).
 * `fn` is either NULL (use this repo's C restatement = cpu_baseline kind "port") or a function pointer taken
 * from oracle/_ref/libaocl_ref.so by the caller (kind "reference"):
 *   codec 0  int    LZ4_compress_default(const char*, char*, int, int) / LZ4_decompress_safe(same)
 *   codec 1  int    snappy_compress(const char*, size_t, char*, size_t*) / snappy_uncompress(same)   (snappy-c.h)
 *   codec 2  size_t ZSTD_compress(void*, size_t, const void*, size_t, int) / ZSTD_decompress(void*, size_t, const void*, size_t)
 */
#include "oracle.h"
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef int (*lz4_fn)(const char*, char*, int, int);
typedef int (*snappy_fn)(const char*, size_t, char*, size_t*);
typedef size_t (*zstd_c_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_d_fn)(void*, size_t, const void*, size_t);

/* Zstd level of the reference function in the drivers below (bench.py's zstd3 line: ZSTD_compress(.., 3)); the C restatement
 * only covers levels 1-2 as a one-shot (levels 3-4 through orc_zstd_compress_cparams), so a level above 2 needs `fn`. */
static int g_zstd_level = 1;
void orc_set_zstd_level(int level) { g_zstd_level = level; }

static int64_t comp_one(int codec, void* fn, const uint8_t* s, size_t n, uint8_t* d, size_t cap)
{
    if (codec == 0) return fn ? ((lz4_fn)fn)((const char*)s, (char*)d, (int)n, (int)cap) : orc_lz4_compress(s, (int)n, d, (int)cap);
    if (codec == 1) {
        if (fn) { size_t l = cap; return ((snappy_fn)fn)((const char*)s, n, (char*)d, &l) == 0 ? (int64_t)l : 0; }
        return (int64_t)orc_snappy_compress(s, n, d);
    }
    if (fn) { size_t r = ((zstd_c_fn)fn)(d, cap, s, n, g_zstd_level); return r > cap ? 0 : (int64_t)r; }
    return (int64_t)(g_zstd_level <= 1 ? orc_zstd_compress(s, n, d, cap) : orc_zstd_compress_level(s, n, d, cap, g_zstd_level));
}
static int64_t decomp_one(int codec, void* fn, const uint8_t* s, size_t n, uint8_t* d, size_t cap)
{
    if (codec == 0) return fn ? ((lz4_fn)fn)((const char*)s, (char*)d, (int)n, (int)cap) : orc_lz4_decompress(s, (int)n, d, (int)cap);
    if (codec == 1) {
        if (fn) { size_t l = cap; return ((snappy_fn)fn)((const char*)s, n, (char*)d, &l) == 0 ? (int64_t)l : -1; }
        { size_t l = 0; if (!orc_snappy_uncompressed_length(s, n, &l)) return -1; return orc_snappy_uncompress(s, n, d, cap) ? (int64_t)l : -1; }
    }
    if (fn) { size_t r = ((zstd_d_fn)fn)(d, cap, s, n); return r > cap ? -1 : (int64_t)r; }
    { size_t r = orc_zstd_decompress(s, n, d, cap); return r == (size_t)-1 ? -1 : (int64_t)r; }
}

int orc_compress_chunks_fn(int codec, void* fn, const uint8_t* src, size_t n, size_t chunk,
                           uint8_t* dst, size_t slot, int32_t* out_len, int threads)
{
    const long nchunks = n ? (long)((n + chunk - 1) / chunk) : 1;
    int fails = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+:fails)
    for (long i = 0; i < nchunks; i++) {
        const size_t o = (size_t)i * chunk;
        const size_t len = (n - o < chunk) ? n - o : chunk;
        const int64_t r = comp_one(codec, fn, src + o, len, dst + (size_t)i * slot, slot);
        out_len[i] = (int32_t)r;
        if (r <= 0) fails++;
    }
    return fails;
}

int orc_decompress_chunks_fn(int codec, void* fn, const uint8_t* src, size_t slot, const int32_t* in_len, size_t nchunks,
                             uint8_t* dst, size_t chunk, size_t n, int threads)
{
    int fails = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+:fails)
    for (long i = 0; i < (long)nchunks; i++) {
        const size_t o = (size_t)i * chunk;
        const size_t len = (n - o < chunk) ? n - o : chunk;
        const int64_t r = decomp_one(codec, fn, src + (size_t)i * slot, (size_t)in_len[i], dst + o, len);
        if (r != (int64_t)len) fails++;
    }
    return fails;
}

int orc_compress_chunks(int codec, const uint8_t* src, size_t n, size_t chunk, uint8_t* dst, size_t slot, uint32_t* out_len, int threads)
{ return orc_compress_chunks_fn(codec, NULL, src, n, chunk, dst, slot, (int32_t*)out_len, threads); }
int orc_decompress_chunks(int codec, const uint8_t* src, size_t slot, const uint32_t* in_len, size_t nchunks,
                          uint8_t* dst, size_t chunk, size_t n, int threads)
{ return orc_decompress_chunks_fn(codec, NULL, src, slot, (const int32_t*)in_len, nchunks, dst, chunk, n, threads); }

/* cpu_baseline timing (bench.py): the timing convention of the reference's own bench (test/codec_bench.c:833-921:
 * `iterations` passes, best and average of the per-pass time; api/api.cpp:58-75: wall clock around the codec call),
 * applied to the chunk decomposition above.
 *   - a private copy of the sample and both output buffers are FIRST-TOUCHED by the thread that will work on them
 *     (schedule(static,1) everywhere, so chunk i always belongs to thread i % threads): no page faults and no
 *     remote-NUMA traffic inside the timed passes;
 *   - one untimed warm-up pass (code, tables, branch predictors, thread pool), then `reps` timed passes of the whole
 *     compress loop and of the whole decompress loop; tc[r] / td[r] = seconds of pass r.
 * Returns failures (compress <= 0, decode size mismatch, or round-trip bytes differ); out_len[] = sizes of the chunks. */
int orc_bench_chunks(int codec, void* fc, void* fd, const uint8_t* sample, size_t n, size_t chunk, size_t slot,
                     int32_t* out_len, int threads, int reps, double* tc, double* td)
{
    const long nchunks = n ? (long)((n + chunk - 1) / chunk) : 1;
    if (threads < 1) threads = 1;
    uint8_t* src = (uint8_t*)malloc(n ? n : 1);
    uint8_t* dst = (uint8_t*)malloc((size_t)nchunks * slot);
    uint8_t* back = (uint8_t*)malloc(n ? n : 1);
    if (!src || !dst || !back) { free(src); free(dst); free(back); return -1; }
    int fails = 0;
#pragma omp parallel for schedule(static, 1) num_threads(threads)
    for (long i = 0; i < nchunks; i++) {                  /* first touch */
        const size_t o = (size_t)i * chunk, len = (n - o < chunk) ? n - o : chunk;
        memcpy(src + o, sample + o, len);
        memset(dst + (size_t)i * slot, 0, slot);
        memset(back + o, 0, len);
    }
    for (int r = -1; r < reps; r++) {                     /* r == -1: warm-up */
        int f = 0;
        double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static, 1) num_threads(threads) reduction(+:f)
        for (long i = 0; i < nchunks; i++) {
            const size_t o = (size_t)i * chunk, len = (n - o < chunk) ? n - o : chunk;
            const int64_t c = comp_one(codec, fc, src + o, len, dst + (size_t)i * slot, slot);
            out_len[i] = (int32_t)c;
            if (c <= 0) f++;
        }
        double t1 = omp_get_wtime();
#pragma omp parallel for schedule(static, 1) num_threads(threads) reduction(+:f)
        for (long i = 0; i < nchunks; i++) {
            const size_t o = (size_t)i * chunk, len = (n - o < chunk) ? n - o : chunk;
            if (out_len[i] <= 0) continue;
            const int64_t d = decomp_one(codec, fd, dst + (size_t)i * slot, (size_t)out_len[i], back + o, len);
            if (d != (int64_t)len) f++;
        }
        double t2 = omp_get_wtime();
        if (r >= 0) { tc[r] = t1 - t0; td[r] = t2 - t1; }
        fails += f;
    }
    if (memcmp(back, src, n) != 0) fails++;
    free(src); free(dst); free(back);
    return fails;
}
